"""Wave-per-pair kernels: time against waves per pair for several pair counts and lengths: python tools/coop_waves_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
cases = [(n, 300, 1000, sa.TRIANGULAR) for n in (8, 16, 23, 32, 46, 64, 91)] + \
        [(n, 2000, 4096, sa.ADJACENT) for n in (17, 65, 257, 513, 1025)] + [(n, 100, 300, sa.TRIANGULAR) for n in (16, 32, 64)]
for n, lo, hi, schedule in cases:
    seqs, _ = synth.generate(n, lo, hi, 2)
    al = sa.Aligner(max_seq_len=hi)
    al.upload(seqs)
    rows = al.rows(schedule)
    pairs = al.pair_count(schedule, 0, rows)
    out = []
    for waves in (1, 2, 4, 8, 16):
        al.set_cooperative(1)
        al.set_cooperative_waves(waves)
        al.prepare_resident(schedule, 0, rows)
        al.run_resident()
        out.append("%d:%.3f" % (waves, min(al.run_resident() for _ in range(3))))
    al.set_cooperative(-1)
    al.prepare_resident(schedule, 0, rows)
    al.run_resident()
    lane = min(al.run_resident() for _ in range(2))
    al.set_cooperative(0)
    al.set_cooperative_waves(0)
    al.prepare_resident(schedule, 0, rows)
    al.run_resident()
    print("%5d pairs len %d-%d  ms by waves %s  pair-per-lane %.3f  library: %s %.3f" % (pairs, lo, hi, " ".join(out), lane, al.last_kernel(), al.run_resident()), flush=True)
    al.close()
