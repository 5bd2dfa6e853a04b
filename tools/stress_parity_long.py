"""Randomised parity sweep of the long-sequence kernels (run on a GPU box; not part of the test suite: it takes minutes).
python tools/stress_parity_long.py [first_seed] [count]
Each case draws a table that keeps the packed kernels in range (shifted scores 1..59: half-precision or unsigned
maxima, bases moved every 16, 8 or 4 rows), a gap penalty, an alphabet of 2..20
letters (small ones give long runs of gaps and many ties), 70-260 sequences and a wide length range (65 .. 1500: pairs
whose lengths differ by a factor of ten have runs of LEFT or UP moves hundreds of columns long, which cross several
64-column strips in one row); every record of the all-vs-all schedule is compared between the packed and the 32-bit
kernel and the wave-per-pair kernels (one wave, a workgroup of waves), and a sample of pairs with the checker."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import sequencealigner_amd as sa

ALPHABET = "ARNDCQEGHILKMFPSTWYV"
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    rng = random.Random(9000 + seed)
    gap = -rng.randint(1, 8)
    top = rng.choice([24, 28, 50, 59])                                        # S' = S - 2 gap <= top: the kernels' two forms
    hi_s = rng.randint(2, top + 2 * gap if top + 2 * gap >= 2 else 2)
    lo_s = max(1 + 2 * gap, hi_s - rng.randint(1, 20))                       # S' >= 1
    matrix = [rng.randint(lo_s, hi_s) for _ in range(400)]
    letters = ALPHABET[: rng.choice([2, 3, 4, 8, 20, 20])]
    n = rng.choice([70, 130, 260])
    lo = rng.choice([65, 65, 100, 300])
    hi = rng.choice([130, 400, 700, 1500])
    if hi < lo:
        lo, hi = 65, hi
    # a few sequences far longer or shorter than the rest
    lens = [rng.randint(lo, hi) for _ in range(n)]
    for _ in range(4):
        lens[rng.randrange(n)] = rng.choice([65, 66, 127, 128, 129, hi, min(2048, 2 * hi)])
    seqs = ["".join(rng.choice(letters) for _ in range(L)) for L in lens]
    al = sa.Aligner(gap=gap, matrix=matrix, max_seq_len=max(lens))
    al.set_cooperative(-1)          # pair-per-lane kernels, whatever the pair count
    al.upload(seqs)
    paired = al.align(sa.TRIANGULAR)
    kernel = al.last_kernel()
    al.set_pairing(False)
    unpaired = al.align(sa.TRIANGULAR)
    kernel2 = al.last_kernel()
    same = np.array_equal(paired, unpaired)
    # the wave-per-pair kernel and a workgroup of waves per pair: the same bytes again
    al.set_cooperative(1)
    coop_ok = True
    for waves in (1, rng.choice([2, 4, 8, 16]), 0):
        al.set_cooperative_waves(waves)
        if not np.array_equal(al.align(sa.TRIANGULAR), unpaired):
            coop_ok = False
            print("  wave-per-pair kernel differs", seed, waves, al.last_kernel(), flush=True)
    al.set_cooperative(-1)
    same = same and coop_ok
    ok = True
    for _ in range(60):
        i = rng.randrange(n - 1)
        j = rng.randrange(i + 1, n)
        k = al.pair_index(sa.TRIANGULAR, i) + j - i - 1
        o1, o2, r = oracle_lib.align(seqs[i], seqs[j], gap=gap, matrix=matrix)
        a1, a2, st = al.expand(paired[k], i, j)
        if (a1, a2, st.score, st.matches, st.gaps) != (o1, o2, r.score, r.matches, r.gaps):
            ok = False
            print("  mismatch", seed, i, j, len(seqs[i]), len(seqs[j]), flush=True)
            break
    bad += (not same) or (not ok)
    print(f"seed {seed}: n {n} len {min(lens)}-{max(lens)} letters {len(letters)} gap {gap} S {lo_s}..{hi_s} {kernel} / {kernel2} agree {same} checker {ok}", flush=True)
    al.close()
print("FAILED" if bad else "all cases agree", bad)
sys.exit(1 if bad else 0)
