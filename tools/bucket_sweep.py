"""Length-bucket width sweep on a protein set (kernel time, records resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seqs, _ = synth.generate(n, 300, 1000, 2)
al = sa.Aligner(max_seq_len=1000)
al.upload(seqs)
rows = al.rows(sa.TRIANGULAR)
cells = al.cell_count(sa.TRIANGULAR, 0, rows)
for w in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,64,32,16,8,4,2,1".split(","))]:
    al.set_bucket_width(w)
    al.prepare_resident(sa.TRIANGULAR, 0, rows)
    al.run_resident()
    ms = min(al.run_resident() for _ in range(2))
    items = al.work_items(sa.TRIANGULAR, 0, rows)
    paired = int((items[:, 5] != 0xFFFFFFFF).sum())
    print(w, round(ms, 2), "ms", round(cells / ms / 1e6), "GCUPS", al.last_kernel(), "items", len(items), "paired", paired, flush=True)
