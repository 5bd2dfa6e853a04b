"""Phase split of the long-sequence kernel's share of a mixed set: python tools/mixed_phase.py npep extra lo hi"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
npep, extra, lo, hi = (int(x) for x in sys.argv[1:5])
seqs, _ = synth.generate(npep, 6, 50, 3)
more, _ = synth.generate(extra, lo, hi, 5)
step = len(seqs) // extra
for k, s in enumerate(more):
    seqs.insert(k * step + k, s)
al = sa.Aligner(max_seq_len=hi)
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
al.prepare_resident(sa.TRIANGULAR, 0, rows)
al.run_resident()
ms = al.run_resident()
al.phase_cycles(True)
al.run_resident()
t = al.phase_cycles(False)
tot = float(sum(t)) or 1.0
items = al.work_items(sa.TRIANGULAR, 0, rows)
print("%s %.1f ms; %d items; long kernel's phases %s" % (al.last_kernel(), ms, len(items),
      " ".join("%d:%.1f%%" % (k, 100.0 * v / tot) for k, v in enumerate(t) if v)))
