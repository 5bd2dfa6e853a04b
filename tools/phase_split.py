"""Phase split of the peptide kernel on the bench workload (in-kernel s_memtime timers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seqs, _ = synth.generate(n, 6, 50, 1)
al = sa.Aligner()
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
al.prepare_resident(sa.TRIANGULAR, 0, rows)
al.run_resident()
al.phase_cycles(True)
ms = al.run_resident()
tot = al.phase_cycles(False)
names = ["fetch", "profile", "operands", "fill", "traceback", "records"]
s = sum(tot[:6])
print(al.last_kernel(), "timed launch", round(ms, 2), "ms")
for k in range(6):
    print(f"{names[k]:10s} {tot[k] / s * 100:5.1f} %")
