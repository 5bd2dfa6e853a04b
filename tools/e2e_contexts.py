"""Mega data set end to end with 1, 2, 3 and 4 contexts on one card: python tools/e2e_contexts.py"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequencealigner_amd as sa
from sequencealigner_amd import synth
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "mega.csv")
synth.make_mega(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "avpdb.csv"), path)
for ndev, write in ((1, 1), (2, 1), (3, 1), (4, 1), (1, 0), (2, 0), (3, 0), (4, 0)):
    out = os.path.join(d, "out.csv") if write else ""
    best = None
    for rep in range(3):
        st = sa.default_settings()
        st.num_devices = ndev
        st.force_devices = 1
        for k in range(ndev):
            st.devices[k] = 0
        rep_ = sa.run_csv(path, out, sa.ADJACENT, settings=st)
        if best is None or rep_.seconds < best.seconds:
            best = rep_
    print(ndev, "contexts,", "write" if write else "no write:", round(best.seconds, 4), "s; write per device", [round(x, 3) for x in best.device_write[:ndev]],
          "wait", [round(x, 3) for x in best.device_wait[:ndev]], "submit", [round(x, 3) for x in best.device_submit[:ndev]],
          "ingest", round(best.seconds_ingest, 3), "upload", round(best.seconds_upload, 3), flush=True)
    if write:
        os.unlink(out)
