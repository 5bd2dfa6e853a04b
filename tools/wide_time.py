"""Long-sequence kernels by gap penalty: the packed kernels' two forms and the 32-bit kernel on the first N synthetic
proteins (BLOSUM50: gap -4 keeps the half-precision maximum, -8 needs the unsigned one).  python tools/wide_time.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seqs, _ = synth.generate(n, 300, 1000, 2)
for gap in (-4, -6, -8, -12):
    al = sa.Aligner(gap=gap, max_seq_len=1000)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    cells = al.cell_count(sa.TRIANGULAR, 0, rows)
    out = []
    for pairing in (True, False):
        al.set_pairing(pairing)
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
        al.run_resident()
        ms = min(al.run_resident() for _ in range(2))
        out.append("%s %.1f ms %.0f GCUPS" % (al.last_kernel(), ms, cells / ms / 1e6))
    print("gap %d: %s" % (gap, " | ".join(out)), flush=True)
    al.close()
