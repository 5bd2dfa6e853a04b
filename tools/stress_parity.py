"""Randomised parity sweep, paired peptide kernel against the CPU checker (run on a GPU box; not part of the test suite:
it takes minutes).  python tools/stress_parity.py [first_seed] [count]
Each case draws a table that keeps the packed kernel in range (shifted scores 1..63: half-precision or unsigned maxima), a gap penalty, an alphabet of 2..20
letters (small ones give long runs of gaps and many ties), a number of sequences (few: wide buckets, partners of
different lengths; many: exact-length buckets) and a length range; every record of the all-vs-all schedule is compared
between the packed and the 32-bit kernel, and a sample of rows with the checker."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import sequencealigner_amd as sa

ALPHABET = "ARNDCQEGHILKMFPSTWYV"
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    rng = random.Random(5000 + seed)
    gap = -rng.randint(1, 8)
    top = rng.choice([24, 30, 63])                                            # S' = S - 2 gap <= top: the kernel's two forms
    hi_s = rng.randint(2, top + 2 * gap if top + 2 * gap >= 2 else 2)
    lo_s = max(1 + 2 * gap, hi_s - rng.randint(1, 20))                       # S' >= 1
    matrix = [rng.randint(lo_s, hi_s) for _ in range(400)]
    letters = ALPHABET[: rng.choice([2, 3, 4, 8, 20, 20])]
    n = rng.choice([70, 130, 300, 900, 2500])
    lo = rng.randint(1, 40)
    hi = rng.randint(lo, 64)
    seqs = ["".join(rng.choice(letters) for _ in range(rng.randint(lo, hi))) for _ in range(n)]
    al = sa.Aligner(gap=gap, matrix=matrix, max_seq_len=64)
    al.upload(seqs)
    paired = al.align(sa.TRIANGULAR)
    kernel = al.last_kernel()
    al.set_pairing(False)
    unpaired = al.align(sa.TRIANGULAR)
    same = np.array_equal(paired, unpaired)
    rows = [0, n - 2] + [rng.randrange(n - 1) for _ in range(6)]
    ok = True
    for i in rows:
        base = al.pair_index(sa.TRIANGULAR, i)
        for j in range(i + 1, min(n, i + 1 + 400)):
            o1, o2, r = oracle_lib.align(seqs[i], seqs[j], gap=gap, matrix=matrix)
            a1, a2, st = al.expand(paired[base + j - i - 1], i, j)
            if (a1, a2, st.score, st.matches, st.gaps) != (o1, o2, r.score, r.matches, r.gaps):
                ok = False
                print("  mismatch", seed, i, j, seqs[i], seqs[j], flush=True)
                break
        if not ok:
            break
    bad += (not same) or (not ok)
    print(f"seed {seed}: n {n} len {lo}-{hi} letters {len(letters)} gap {gap} S {lo_s}..{hi_s} {kernel} kernels agree {same} checker {ok}", flush=True)
    al.close()
print("FAILED" if bad else "all cases agree", bad)
sys.exit(1 if bad else 0)
