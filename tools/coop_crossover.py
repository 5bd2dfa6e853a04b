"""Where the wave-per-pair kernel stops paying: proteins all-vs-all at growing counts, both kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
for n in (182, 260, 362, 512, 724, 1024):
    seqs, _ = synth.generate(n, 300, 1000, 2)
    al = sa.Aligner(max_seq_len=1000)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    line = [n, al.pair_count(sa.TRIANGULAR, 0, rows)]
    for mode in (1, -1):
        al.set_cooperative(mode)
        al.prepare_resident(sa.TRIANGULAR, 0, rows)
        al.run_resident()
        line.append(round(min(al.run_resident() for _ in range(3)), 2))
    print(*line, flush=True)
    al.close()
