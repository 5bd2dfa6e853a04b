import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lo, hi, seed = (int(x) for x in os.environ.get('SA_LENS', '6,50,1').split(','))
t = time.time(); seqs, _ = synth.generate(n, lo, hi, seed); print('gen', time.time()-t, flush=True)
al = sa.Aligner(max_seq_len=max(64, hi))
if os.environ.get("SA_PAIRING"): al.set_pairing(int(os.environ["SA_PAIRING"]))
t = time.time(); al.upload(seqs); print('upload', time.time()-t, flush=True)
rows = al.rows(sa.TRIANGULAR)
pairs = al.pair_count(sa.TRIANGULAR, 0, rows); cells = al.cell_count(sa.TRIANGULAR, 0, rows)
print('pairs', pairs, 'cells', cells, 'stride', al.stride, 'out GB', pairs*al.stride/1e9, flush=True)
for tiles in [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['8'])]:
    al.set_tuning(tiles_per_item=tiles, wg_per_cu=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    t = time.time(); al.prepare_resident(sa.TRIANGULAR, 0, rows); print('prepare', time.time()-t, flush=True)
    for it in range(3):
        ms = al.run_resident()
        print(f'tiles/item={tiles} run {it}: {ms:.2f} ms  {cells/ms/1e6:.1f} GCUPS  {pairs/ms/1e3:.2f} Mpairs/s', flush=True)
    al.phase_cycles(True); ms = al.run_resident(); ph = al.phase_cycles(False)
    tot = sum(ph) or 1
    print(f'timed run {ms:.2f} ms; phase share: ' + ' '.join(f'{n}={100*v/tot:.1f}%' for n, v in zip(['fetch','profile','operands','fill','traceback','store'], ph)), flush=True)
