"""Wave-per-pair kernel timings on the two few-pairs cases of bench.py (for A/B runs with SA_HIP_LIBRARY)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
for n, lo, hi, seed, schedule in ((65, 4096, 4096, 5, sa.ADJACENT), (150, 300, 1000, 2, sa.TRIANGULAR), (362, 300, 1000, 2, sa.TRIANGULAR)):
    seqs, _ = synth.generate(n, lo, hi, seed)
    al = sa.Aligner(max_seq_len=hi)
    al.upload(seqs)
    al.set_cooperative(1)
    rows = al.rows(schedule)
    for waves in (1, 2, 4, 8, 16, 0):
        al.set_cooperative_waves(waves)
        al.prepare_resident(schedule, 0, rows)
        al.run_resident()
        print(n, hi, "waves", waves, round(min(al.run_resident() for _ in range(3)), 3), "ms", al.last_kernel(), flush=True)
    al.close()
