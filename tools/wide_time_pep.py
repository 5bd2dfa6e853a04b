"""Peptide kernels by gap penalty: the packed kernel's two forms and the 32-bit kernel on 20 000 synthetic peptides.
python tools/wide_time_pep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
seqs, _ = synth.generate(20000, 6, 50, 1)
for gap in (-4, -10, -14, -24):
    al = sa.Aligner(gap=gap, max_seq_len=64)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    cells = al.cell_count(sa.TRIANGULAR, 0, rows)
    out = []
    for pairing in (True, False):
        al.set_pairing(pairing)
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
        al.run_resident()
        ms = min(al.run_resident() for _ in range(3))
        out.append("%s %.2f ms %.0f GCUPS" % (al.last_kernel(), ms, cells / ms / 1e6))
    print("gap %d: %s" % (gap, " | ".join(out)), flush=True)
    al.close()
