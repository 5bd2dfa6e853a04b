#!/usr/bin/env python3
"""Writes the layout fixtures: tests/golden/layout_in.csv and the outputs of the REFERENCE itself
(oracle/_ref/ref_layout_a, ref_layout_b: its main() compiled from its sources with other layout
settings, see oracle/Makefile) on that input.  Run from the repository root after `make -C oracle ref`.
The fixtures are data: a three-column input and what the reference writes for it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sequencealigner_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
seqs, labels = synth.generate(90, 1, 40, seed=2024)
src = os.path.join(GOLDEN, "layout_in.csv")
with open(src, "w", newline="") as f:
    f.write("id,sequence,label\n")
    for k, (s, l) in enumerate(zip(seqs, labels)):
        s = s.decode() if isinstance(s, bytes) else s
        f.write(f"p{k * 7 % 1000},{s},{l}\n")
for name in ("a", "b"):
    out = os.path.join(GOLDEN, f"layout_{name}_out.csv")
    env = dict(os.environ, SA_REF_INPUT=src, SA_REF_OUTPUT=out)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", f"ref_layout_{name}")], check=True, env=env,
                   stdout=subprocess.DEVNULL)
    print(out, os.path.getsize(out), "bytes")
