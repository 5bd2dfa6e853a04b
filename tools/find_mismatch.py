"""Aligns a CSV's sequences all-vs-all on the GPU and lists the pairs whose record differs from the CPU checker."""
import sys, os, csv
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle_lib
import sequencealigner_amd as sa

path = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/avpdb.csv"
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = list(csv.reader(open(path)))
col = rows[0].index("sequence") if "sequence" in rows[0] else 0
seqs = [r[col] for r in rows[1:]][:limit]
al = sa.Aligner(max_seq_len=64); al.upload(seqs)
rec = al.align(sa.TRIANGULAR)
n = len(seqs); k = 0; bad = 0
from collections import Counter
hist = Counter()
for i in range(n - 1):
    for j in range(i + 1, n):
        o1, o2, r = oracle_lib.align(seqs[i], seqs[j])
        try:
            a1, a2, st = al.expand(rec[k], i, j)
            ok = (a1, a2) == (o1, o2) and st.matches == r.matches and st.gaps == r.gaps and st.score == r.score
        except Exception as e:
            ok = False; a1 = a2 = repr(e); st = None
        if not ok:
            bad += 1
            hist[(len(seqs[i]), len(seqs[j]) % 4, (st.matches - r.matches) if st else None, (a1, a2) == (o1, o2))] += 1
            if bad <= 3:
                print("pair", i, j, "len", len(seqs[i]), len(seqs[j]), "record", bytes(rec[k][:8]).hex(), flush=True)
                print("  gpu ", a1, a2, st and (st.score, st.matches, st.gaps)); print("  cpu ", o1, o2, (r.score, r.matches, r.gaps))
        k += 1
print("pairs", k, "bad", bad)
for key, v in sorted(hist.items(), key=lambda kv: -kv[1])[:40]:
    print(key, v)
