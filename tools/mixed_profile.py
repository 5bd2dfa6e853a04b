"""One mixed set under the profiler: python tools/mixed_profile.py [long_lo long_hi]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (80, 120)
seqs, _ = synth.generate(20000, 6, 50, 3)
more, _ = synth.generate(20, lo, hi, 5)
step = len(seqs) // 20
for k, s in enumerate(more):
    seqs.insert(k * step + k, s)
al = sa.Aligner(max_seq_len=hi)
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
al.prepare_resident(sa.TRIANGULAR, 0, rows)
for _ in range(3):
    ms = al.run_resident()
print(al.last_kernel(), ms)
