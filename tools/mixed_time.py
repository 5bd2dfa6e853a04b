"""Peptide sets with a few long sequences among them, with the peptide tiles on the peptide kernel and without:
python tools/mixed_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
for npep, extra, lo, hi in ((20000, 0, 0, 0), (20000, 20, 80, 120), (20000, 600, 65, 200), (8000, 0, 0, 0), (8000, 20, 80, 120), (8000, 20, 300, 1000), (8000, 200, 300, 1000)):
    seqs, _ = synth.generate(npep, 6, 50, 3)
    if extra:
        more, _ = synth.generate(extra, lo, hi, 5)
        step = len(seqs) // extra
        for k, s in enumerate(more):
            seqs.insert(k * step + k, s)
    al = sa.Aligner(max_seq_len=max(64, hi))
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    cells = al.cell_count(sa.TRIANGULAR, 0, rows)
    out = []
    digests = []
    pairs = al.pair_count(sa.TRIANGULAR, 0, rows)
    for tiles in ((True, False) if extra else (True,)):
        al.set_peptide_tiles(tiles)
        try:
            al.prepare_resident(sa.TRIANGULAR, 0, rows)
        except sa.SaError as e:      # the records of every pair at the stride of the longest sequence do not fit the card
            out.append(str(e))
            break
        al.run_resident()
        ms = min(al.run_resident() for _ in range(2))
        digests.append(tuple(al.digest_resident(0, pairs)))
        out.append("%s %.1f ms %.0f GCUPS" % (al.last_kernel(), ms, cells / ms / 1e6))
    if len(digests) == 2:
        out.append("digests of all %d records %s" % (pairs, "equal" if digests[0] == digests[1] else "DIFFER"))
    print("%d peptides + %d of %d-%d residues, %d-byte records, cells %.2e: %s" % (npep, extra, lo, hi, al.stride, cells, " | ".join(out)), flush=True)
    al.close()
