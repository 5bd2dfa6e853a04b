"""The bench workload cut into N row ranges of equal cell count, each range timed alone on one GPU: what an N-GPU run's
ranks would take if nothing but the kernels mattered.  python tools/shard_times.py [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
for name, (n, lo, hi, seed) in (("peptides50k", (50000, 6, 50, 1)), ("proteins10k", (10000, 300, 1000, 2))):
    seqs, _ = synth.generate(n, lo, hi, seed)
    al = sa.Aligner(max_seq_len=max(64, hi))
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    al.prepare_resident(sa.TRIANGULAR, 0, rows)
    al.run_resident()
    whole = min(al.run_resident() for _ in range(2))
    print(name, "whole", round(whole, 2), "ms", flush=True)
    for parts in [int(x) for x in sys.argv[1:]] or [2, 4, 8]:
        b = al.shard_rows(sa.TRIANGULAR, parts)
        ms = []
        for r in range(parts):
            al.prepare_resident(sa.TRIANGULAR, b[r], b[r + 1])
            al.run_resident()
            ms.append(min(al.run_resident() for _ in range(2)))
        print(name, parts, "ranges:", [round(x, 2) for x in ms], "slowest x N / whole =", round(max(ms) * parts / whole, 3), flush=True)
    al.close()
