"""Times the long-sequence kernels on the first N synthetic proteins: python tools/ckpt_time.py [n] [ck,ck,...] [phases]"""
import sys
sys.path.insert(0, ".")
import sequencealigner_amd as sa
from sequencealigner_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 32, 64]
import os
lo, hi = int(os.environ.get("SA_LEN_LO", 300)), int(os.environ.get("SA_LEN_HI", 1000))
seqs, _ = synth.generate(n, lo, hi, 2)
al = sa.Aligner(max_seq_len=hi)
al.upload(seqs)
import os
if os.environ.get("SA_WG_PER_CU"):
    al.set_tuning(wg_per_cu=int(os.environ["SA_WG_PER_CU"]))
rows = al.rows(sa.TRIANGULAR)
cells = al.cell_count(sa.TRIANGULAR, 0, rows)
for ck in cks:
    al.set_long_checkpoint_rows(ck)
    al.prepare_resident(sa.TRIANGULAR, 0, rows)
    al.run_resident()
    ms = [al.run_resident() for _ in range(4)]
    d = al.digest_resident(0, al.pair_count(sa.TRIANGULAR, 0, rows))
    line = "n=%d ck=%d %s %.2f ms (%.2f) %.0f GCUPS digest %016x" % (n, ck, al.last_kernel(), min(ms), sum(ms) / len(ms), cells / min(ms) / 1e6, d[0])
    if len(sys.argv) > 3:
        al.phase_cycles(True)
        al.run_resident()
        ph = al.phase_cycles(False)
        tot = sum(ph) or 1
        tot = sum(ph[:6]) or 1
        line += " phases " + " ".join("%.1f" % (100.0 * p / tot) for p in ph[:6]) + " rows refilled %d walk rows %d" % (ph[6], ph[7])
    print(line, flush=True)
