"""The headline workload's kernel time, for A/B runs with SA_HIP_LIBRARY: python tools/pep_ab.py [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
seqs, _ = synth.generate(50000, 6, 50, 1)
al = sa.Aligner()
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
al.prepare_resident(sa.TRIANGULAR, 0, rows)
al.run_resident()
ms = [al.run_resident() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5)]
print(os.environ.get("SA_HIP_LIBRARY", "tree"), al.last_kernel(), "min %.2f mean %.2f ms" % (min(ms), sum(ms) / len(ms)), flush=True)
