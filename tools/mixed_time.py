"""A peptide set with a few long sequences among it (the whole set then runs on a long-sequence kernel):
python tools/mixed_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
pep, _ = synth.generate(20000, 6, 50, 3)
for extra, lo, hi in ((0, 0, 0), (20, 80, 120), (20, 300, 1000), (200, 300, 1000)):
    seqs = list(pep)
    if extra:
        more, _ = synth.generate(extra, lo, hi, 5)
        step = len(seqs) // extra
        for k, s in enumerate(more):
            seqs.insert(k * step + k, s)
    al = sa.Aligner(max_seq_len=max(64, hi))
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    cells = al.cell_count(sa.TRIANGULAR, 0, rows)
    try:
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
    except sa.SaError as e:      # the records of every pair at the stride of the longest sequence do not fit the card
        print("20000 peptides + %d of %d-%d residues: %d-byte records, %s" % (extra, lo, hi, al.stride, e), flush=True)
        al.close()
        continue
    al.run_resident()
    ms = min(al.run_resident() for _ in range(2))
    print("20000 peptides + %d of %d-%d residues: cells %.2e, %s %.1f ms %.0f GCUPS" % (extra, lo, hi, cells, al.last_kernel(), ms, cells / ms / 1e6), flush=True)
    al.close()
