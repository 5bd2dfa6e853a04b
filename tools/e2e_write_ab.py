"""Mega data set end to end with one context, output on a tmpfs and in /tmp, best and median of 7 (for A/B runs with
SA_HIP_LIBRARY): python tools/e2e_write_ab.py"""
import os, sys, tempfile, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "mega.csv")
synth.make_mega(os.path.join(root, "tests", "golden", "avpdb.csv"), path)
for outdir in (d, tempfile.mkdtemp(dir="/tmp")):
    out = os.path.join(outdir, "out.csv")
    secs, writes = [], []
    for rep in range(7):
        r = sa.run_csv(path, out, sa.ADJACENT)
        secs.append(r.seconds); writes.append(r.device_write[0])
    h = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]
    os.unlink(out)
    secs.sort(); writes.sort()
    print(os.environ.get("SA_HIP_LIBRARY", "tree"), outdir[:8], "seconds best %.4f median %.4f; write best %.4f median %.4f; sha256 %s" % (secs[0], secs[3], writes[0], writes[3], h), flush=True)
