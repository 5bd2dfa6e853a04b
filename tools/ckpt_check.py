"""Development check of the checkpointing long-sequence kernel against the direction-matrix kernel (every record) and
the CPU checker (sampled pairs), then timings of both on the first 2000 synthetic proteins."""
import random
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle_lib
import sequencealigner_amd as sa
from sequencealigner_amd import synth

ALPHABET = "ARNDCQEGHILKMFPSTWYV"


def check(name, seqs, max_len, oracle_pairs=20):
    al = sa.Aligner(max_seq_len=max_len)
    al.upload(seqs)
    al.set_cooperative(-1)
    n = len(seqs)
    recs = {}
    for rows in (0, 32, 64):
        al.set_long_checkpoint_rows(rows)
        recs[rows] = al.align(sa.TRIANGULAR)
        print(name, rows, al.last_kernel(), flush=True)
    for rows in (32, 64):
        if not np.array_equal(recs[0], recs[rows]):
            bad = np.nonzero((recs[0] != recs[rows]).any(axis=1))[0]
            print("  MISMATCH ck=%d: %d of %d records differ; first %s" % (rows, len(bad), len(recs[0]), bad[:10]), flush=True)
            k = int(bad[0])
            d = np.nonzero(recs[0][k] != recs[rows][k])[0]
            print("   record", k, "bytes", d[:20], recs[0][k][d[:20]], recs[rows][k][d[:20]])
            return False
    rng = random.Random(1)
    for _ in range(oracle_pairs):
        i = rng.randrange(n - 1)
        j = rng.randrange(i + 1, n)
        k = al.pair_index(sa.TRIANGULAR, i) + (j - i - 1)
        o1, o2, r = oracle_lib.align(al.seqs[i], al.seqs[j])
        a1, a2, st = al.expand(recs[32][k], i, j)
        assert (a1, a2) == (o1, o2) and st.score == r.score, (i, j)
    al.close()
    print("  ok", name, flush=True)
    return True


def main():
    ok = True
    rng = random.Random(7)
    seqs, _ = synth.generate(150, 300, 1000, 42)
    ok &= check("proteins150", seqs, 1000)
    seqs = ["".join(rng.choice(ALPHABET) for _ in range(rng.randint(65, 400))) for _ in range(260)]
    ok &= check("len65-400", seqs, 400)
    seqs = ["".join(rng.choice("AG") for _ in range(rng.randint(130, 330))) for _ in range(120)]
    ok &= check("two letters", seqs, 330)
    seqs = ["".join(rng.choice(ALPHABET) for _ in range(rng.choice([65, 66, 127, 128, 129, 191, 192, 193, 2000, 70]))) for _ in range(140)]
    ok &= check("ragged", seqs, 2000, oracle_pairs=8)
    if not ok:
        sys.exit(1)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        seqs, _ = synth.generate(2000, 300, 1000, 2)
        al = sa.Aligner(max_seq_len=1000)
        al.upload(seqs)
        rows = al.rows(sa.TRIANGULAR)
        cells = al.cell_count(sa.TRIANGULAR, 0, rows)
        for ck in (0, 32, 64):
            al.set_long_checkpoint_rows(ck)
            al.prepare_resident(sa.TRIANGULAR, 0, rows)
            al.run_resident()
            ms = [al.run_resident() for _ in range(3)]
            d = al.digest_resident(0, al.pair_count(sa.TRIANGULAR, 0, rows))
            print("proteins2k ck=%d %s %.2f ms %.0f GCUPS digest %x" % (ck, al.last_kernel(), min(ms), cells / min(ms) / 1e6, d[0]), flush=True)
            al.phase_cycles(True)
            al.run_resident()
            ph = al.phase_cycles(False)
            tot = sum(ph) or 1
            print("   phases", [round(100.0 * p / tot, 1) for p in ph], flush=True)


main()
