"""GPU box check of the long-sequence kernel against the CPU checker (whole-file byte equality)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
import oracle_lib

ok = True
cases = [(40, 65, 200, 5), (120, 1, 300, 6), (64, 128, 128, 7), (150, 300, 1000, 2)]
if len(sys.argv) > 1: cases = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for n, lo, hi, seed in cases:
    seqs, labels = synth.generate(n, lo, hi, seed)
    synth.write_csv('/tmp/l_in.csv', seqs, labels)
    for sched, name in ((sa.TRIANGULAR, 'triangular'), (sa.ADJACENT, 'adjacent')):
        t = time.time(); st = oracle_lib.run_file('/tmp/l_in.csv', '/tmp/l_or.csv', sched, threads=16); to = time.time() - t
        rep = sa.run_csv('/tmp/l_in.csv', '/tmp/l_gpu.csv', sched, max_seq_len=hi + 1)
        same = open('/tmp/l_or.csv', 'rb').read() == open('/tmp/l_gpu.csv', 'rb').read()
        print(f'n={n} len {lo}-{hi} {name}: pairs {rep.pairs} cells {rep.cells} gpu {rep.seconds:.3f}s ({rep.cells/rep.seconds/1e9:.1f} GCUPS) oracle {to:.2f}s identical: {same}', flush=True)
        ok &= same
sys.exit(0 if ok else 1)
