"""Phase split (shader cycles per phase, summed over workgroups) of a synthetic all-vs-all set:
python tools/phase_split.py n lo hi [checkpoint_rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seqs, _ = synth.generate(n, lo, hi, 3)
al = sa.Aligner(max_seq_len=max(64, hi))
if len(sys.argv) > 4:
    al.set_long_checkpoint_rows(int(sys.argv[4]))
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
cells = al.cell_count(sa.TRIANGULAR, 0, rows)
al.prepare_resident(sa.TRIANGULAR, 0, rows)
al.run_resident()
ms = al.run_resident()
al.phase_cycles(True)
al.run_resident()
t = al.phase_cycles(False)
tot = float(sum(t)) or 1.0
print("%d sequences %d-%d: %s %.1f ms %.0f GCUPS; phases %s" % (n, lo, hi, al.last_kernel(), ms, cells / ms / 1e6,
      " ".join("%d:%.1f%%" % (k, 100.0 * v / tot) for k, v in enumerate(t) if v)))
