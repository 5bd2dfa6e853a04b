"""Randomised parity sweep of sets that hold peptides and long sequences (run on a GPU box; not part of the test suite).
python tools/stress_parity_mixed.py [first_seed] [count]
Each case draws a table (shifted scores 1..63: both forms of the packed peptide cells; the long kernels packed or, beyond
59, with 32-bit cells), a gap penalty, an alphabet of 2..20 letters, 100-1500 peptides of 1-64 residues and 1-60
sequences of 65-900 residues somewhere among them.  Every record of the all-vs-all schedule is compared between the run
with the peptide tiles on the peptide kernel and the run with every tile on the long-sequence kernel, then with the
32-bit kernels, and a sample of pairs of every kind with the checker."""
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
    rng = random.Random(7000 + seed)
    gap = -rng.randint(1, 8)
    top = rng.choice([24, 28, 30, 50, 59, 63])
    hi_s = rng.randint(2, top + 2 * gap if top + 2 * gap >= 2 else 2)
    lo_s = max(1 + 2 * gap, hi_s - rng.randint(1, 20))                       # S' >= 1
    matrix = [rng.randint(lo_s, hi_s) for _ in range(400)]
    letters = ALPHABET[: rng.choice([2, 3, 4, 8, 20, 20])]
    npep = rng.choice([100, 300, 700, 1500])
    plo = rng.choice([1, 6, 30, 60])
    phi = rng.choice([12, 50, 52, 53, 64])
    if phi < plo:
        plo = 1
    lens = [rng.randint(plo, phi) for _ in range(npep)]
    nlong = rng.choice([1, 2, 7, 20, 60])
    llo, lhi = rng.choice([(65, 66), (65, 128), (70, 300), (300, 900)])
    for _ in range(nlong):
        lens.insert(rng.randrange(len(lens) + 1), rng.randint(llo, lhi))
    n = len(lens)
    seqs = ["".join(rng.choice(letters) for _ in range(L)) for L in lens]
    al = sa.Aligner(gap=gap, matrix=matrix, max_seq_len=max(lens))
    al.set_cooperative(-1)          # pair-per-lane kernels, whatever the pair count
    al.upload(seqs)
    both = al.align(sa.TRIANGULAR)
    kernel = al.last_kernel()
    al.set_peptide_tiles(False)
    alone = al.align(sa.TRIANGULAR)
    al.set_pairing(False)
    plain = al.align(sa.TRIANGULAR)
    kernel2 = al.last_kernel()
    same = np.array_equal(both, alone) and np.array_equal(both, plain)
    ok = True
    long_ix = [k for k, L in enumerate(lens) if L > 64]
    sample = []
    for _ in range(60):
        i = rng.randrange(n - 1)
        sample.append((i, rng.randrange(i + 1, n)))
    for i in long_ix[:5]:
        sample += [(min(i, j), max(i, j)) for j in (rng.randrange(n) for _ in range(4)) if j != i]
    for i, j in sample:
        k = al.pair_index(sa.TRIANGULAR, i) + j - i - 1
        o1, o2, r = oracle_lib.align(seqs[i], seqs[j], gap=gap, matrix=matrix)
        a1, a2, st = al.expand(both[k], i, j)
        if (a1, a2, st.score, st.matches, st.gaps) != (o1, o2, r.score, r.matches, r.gaps):
            ok = False
            print("  mismatch", seed, i, j, len(seqs[i]), len(seqs[j]), flush=True)
            break
    bad += (not same) or (not ok)
    print(f"seed {seed}: {npep} peptides {plo}-{phi} + {nlong} of {llo}-{lhi} letters {len(letters)} gap {gap} S {lo_s}..{hi_s} {kernel} / {kernel2} agree {same} checker {ok}", flush=True)
    al.close()
print("FAILED" if bad else "all cases agree", bad)
sys.exit(1 if bad else 0)
