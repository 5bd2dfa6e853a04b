"""Long-sequence kernels on sets between peptides and proteins: python tools/midlen_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
for n, lo, hi in ((30000, 40, 64), (20000, 50, 80), (16000, 65, 128), (10000, 100, 300), (6000, 200, 500), (3000, 600, 1400), (1500, 1500, 3000)):
    seqs, _ = synth.generate(n, lo, hi, 3)
    al = sa.Aligner(max_seq_len=hi)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    cells = al.cell_count(sa.TRIANGULAR, 0, rows)
    out = []
    for ck in (64, 0):
        if hi > 64:
            al.set_long_checkpoint_rows(ck)
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
        al.run_resident()
        ms = min(al.run_resident() for _ in range(2))
        out.append("%s %.1f ms %.0f GCUPS" % (al.last_kernel(), ms, cells / ms / 1e6))
        if hi <= 64:
            break
    print("n %d len %d-%d cells %.2e: %s" % (n, lo, hi, cells, " | ".join(out)), flush=True)
    al.close()
