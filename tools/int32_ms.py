"""Kernel time of the 32-bit (unpaired) peptide kernel on the bench workload: python tools/int32_ms.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
seqs, _ = synth.generate(50000, 6, 50, 1)
al = sa.Aligner()
al.upload(seqs)
al.set_pairing(False)
rows = al.rows(sa.TRIANGULAR)
al.prepare_resident(sa.TRIANGULAR, 0, rows)
al.run_resident()
print(os.environ.get("SA_HIP_LIBRARY", "default"), al.last_kernel(), round(min(al.run_resident() for _ in range(4)), 2), "ms", flush=True)
