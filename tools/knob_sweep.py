"""Sweep of the peptide kernel's launch knobs (tiles per work item, workgroups per CU) on the bench workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seqs, _ = synth.generate(n, 6, 50, 1)
al = sa.Aligner()
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
for waves in (15, 16, 17, 18, 20):
    for tiles in (4, 8, 16, 32, 64):
        al.set_tuning(tiles, waves)
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
        al.run_resident()
        ms = min(al.run_resident() for _ in range(3))
        print(f"waves/CU {waves:2d} tiles/item {tiles:2d}: {ms:7.2f} ms", flush=True)
