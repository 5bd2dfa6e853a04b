"""GPU box check: CSV in -> CSV out through sa_run_csv against the golden hashes."""
import hashlib, os, sys, time
sys.path.insert(0, os.getcwd())
import sequencealigner_amd as sa
from sequencealigner_amd import synth

def sha(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        while True:
            b = f.read(1 << 24)
            if not b: break
            h.update(b)
    return h.hexdigest()

ok = True
rep = sa.run_csv('tests/golden/avpdb.csv', '/tmp/adj.csv', sa.ADJACENT)
same = open('/tmp/adj.csv','rb').read() == open('tests/golden/avpdb_adjacent_results.csv','rb').read()
print('avpdb adjacent: pairs', rep.pairs, 'seconds %.4f' % rep.seconds, 'identical to golden:', same, flush=True); ok &= same
rep = sa.run_csv('tests/golden/avpdb.csv', '/tmp/tri.csv', sa.TRIANGULAR)
h = sha('/tmp/tri.csv'); same = h == '3eb3798aa4d2adff6aabe7bf69f06ef1425d2c25005b456f2a6440eff47e1e08'
print('avpdb triangular: pairs', rep.pairs, 'bytes', rep.bytes_written, 'seconds %.4f' % rep.seconds, 'GCUPS %.1f' % (rep.cells/rep.seconds/1e9), 'sha ok:', same, flush=True); ok &= same
t = time.time(); synth.make_mega('tests/golden/avpdb.csv', '/tmp/mega.csv'); print('mega generated in %.2f s' % (time.time()-t), 'sha ok:', sha('/tmp/mega.csv') == 'bc12a79ad0414d806acad1a4a846fba4e3ba26df1f15b7769161442dd5e70de6', flush=True)
for thr in (16,):
    rep = sa.run_csv('/tmp/mega.csv', '/tmp/mega_out.csv', sa.ADJACENT, format_threads=thr)
    h = sha('/tmp/mega_out.csv'); same = h == '667f3524a9ba31243de55c075165b336010ffeaf76600b39540b54f5dd085f18'
    print('mega adjacent: pairs', rep.pairs, 'bytes', rep.bytes_written, 'seconds %.3f' % rep.seconds, 'Mpairs/s %.2f' % (rep.pairs/rep.seconds/1e6), 'sha ok:', same, 'ingest %.3f upload %.3f wait %.3f format %.3f write %.3f' % (rep.seconds_ingest, rep.seconds_upload, rep.seconds_gpu_wait, rep.seconds_format, rep.seconds_write), flush=True); ok &= same
rep = sa.run_csv('/tmp/mega.csv', None, sa.ADJACENT)
print('mega adjacent no-write: seconds %.3f' % rep.seconds, 'Mpairs/s %.2f' % (rep.pairs/rep.seconds/1e6), flush=True)
sys.exit(0 if ok else 1)
