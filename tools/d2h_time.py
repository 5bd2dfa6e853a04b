"""Times the record path of the CSV driver (no write): python tools/d2h_time.py [n]"""
import os, sys, tempfile
sys.path.insert(0, ".")
import sequencealigner_amd as sa
from sequencealigner_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seqs, labels = synth.generate(n, 6, 50, 1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "p.csv")
synth.write_csv(path, seqs, labels)
for compact in (1, 0, 1, 0):
    rep = sa.run_csv(path, None, sa.TRIANGULAR, compact_records=compact)
    print("compact=%d %.3f s  %.1f GCUPS  %.2f GB of records  %.1f GB/s  wait %.3f" % (compact, rep.seconds, rep.cells / rep.seconds / 1e9,
          rep.record_bytes / 1e9, rep.record_bytes / rep.seconds / 1e9, rep.seconds_gpu_wait), flush=True)
os.unlink(path)
synth.write_csv(path, seqs, labels)
for pps in (1 << 21, 1 << 22, 1 << 23, 1 << 24):
    rep = sa.run_csv(path, None, sa.TRIANGULAR, compact_records=1, pairs_per_submit=pps)
    print("pairs per submission %d: %.3f s  %.1f GB/s  chunks %d  submit %.3f wait %.3f" % (pps, rep.seconds, rep.record_bytes / rep.seconds / 1e9, rep.chunks_done, rep.device_submit[0], rep.device_wait[0]), flush=True)
os.unlink(path)
