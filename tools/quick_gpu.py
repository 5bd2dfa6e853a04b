import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import sequencealigner_amd as sa
import oracle_lib
seqs = [l.split(',')[0] for l in open('tests/golden/avpdb.csv').read().split('\n')[1:] if l]
al = sa.Aligner()
al.upload(seqs)
print('n', len(seqs), 'stride', al.stride, flush=True)
for sched, name, nrows in ((sa.ADJACENT, 'adjacent', None), (sa.TRIANGULAR, 'triangular', 40)):
    r1 = nrows if nrows else al.rows(sched)
    t = time.time(); rec = al.align(sched, 0, r1); print(name, 'records', rec.shape, 'time', time.time()-t, flush=True)
    bad = 0; k = 0
    pairs = [(i, i+1) for i in range(r1)] if sched == sa.ADJACENT else [(i, j) for i in range(r1) for j in range(i+1, len(seqs))]
    for k, (i, j) in enumerate(pairs):
        o1, o2, o = oracle_lib.align(seqs[i], seqs[j])
        try:
            a1, a2, st = al.expand(rec[k], i, j)
            ok = (a1 == o1 and a2 == o2 and st.score == o.score and st.matches == o.matches and st.gaps == o.gaps and st.mismatches == o.mismatches and st.similarity_bp == o.similarity_bp)
        except sa.SaError as e:
            ok = False; a1 = a2 = str(e).encode(); st = None
        if not ok:
            bad += 1
            if bad <= 5:
                print('MISMATCH', name, k, i, j, seqs[i], seqs[j], a1, a2, st and (st.score, st.matches, st.gaps), 'oracle', o1, o2, o.score, o.matches, o.gaps, bytes(rec[k][:24]).hex())
    print(name, 'pairs', len(pairs), 'bad', bad, flush=True)
    if bad: sys.exit(1)
