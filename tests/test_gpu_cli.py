"""The command-line program a user of the reference would run (sa_main stands for the reference's
bin/main, src/main.c:7-134), and the multi-rank launch of bench.py rehearsed on one card.

sa_main is driven the way the reference's GUI drives bin/main: a user.h-style settings file with
INPUT_FILE / OUTPUT_FILE written into it (scripts/config_schema.py:201-241), the program's standard
output scraped for "Alignment time:" (src/main.c:133).
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, sha256_file

pytestmark = pytest.mark.gpu
SA_MAIN = os.path.join(ROOT, "sequencealigner_amd", "sa_main")


def _user_h(path, **defines):
    with open(path, "w") as f:
        f.write("#ifndef USER_H\n#define USER_H\n")
        for k, v in defines.items():
            f.write("#define %s %s\n" % (k, '"%s"' % v if isinstance(v, str) else v))
        f.write("#endif\n")


def _run(args, timeout=300):
    return subprocess.run([SA_MAIN] + args, capture_output=True, text=True, timeout=timeout)


def test_sa_main_with_a_settings_file_reproduces_the_reference_results_file(golden_dir, tmp_path):
    out = tmp_path / "results.csv"
    cfg = tmp_path / "user.h"
    _user_h(cfg, INPUT_FILE=os.path.join(golden_dir, "avpdb.csv"), OUTPUT_FILE=str(out), GAP_PENALTY=-4,
            MAX_SEQ_LEN=64, MODE_WRITE=1, BATCH_SIZE=32768)
    p = _run(["--config", str(cfg)])
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0].startswith("Alignment time: ") and lines[0].endswith(" seconds")
    float(lines[0].split()[2])
    info = json.loads(lines[1])
    assert info["pairs"] == 1041 and info["cells"] == 575742 and info["rows"] == 1042
    assert out.read_bytes() == open(os.path.join(golden_dir, "avpdb_adjacent_results.csv"), "rb").read()


def test_sa_main_triangular_schedule_matches_the_pinned_hash(golden_dir, tmp_path, hashes):
    out = tmp_path / "tri.csv"
    p = _run(["--input", os.path.join(golden_dir, "avpdb.csv"), "--output", str(out), "--schedule", "triangular"])
    assert p.returncode == 0, p.stderr
    assert p.stdout.startswith("Alignment time: ")
    assert json.loads(p.stdout.splitlines()[1])["pairs"] == 542361
    assert sha256_file(out) == hashes["avpdb_triangular_out"]
    # the host formatter gives the same file
    out2 = tmp_path / "tri_host.csv"
    p = _run(["--input", os.path.join(golden_dir, "avpdb.csv"), "--output", str(out2), "--schedule", "triangular",
              "--host-format"])
    assert p.returncode == 0, p.stderr
    assert sha256_file(out2) == hashes["avpdb_triangular_out"]


def test_sa_main_reports_a_bad_input_row_and_exits_with_1(tmp_path):
    src = tmp_path / "bad.csv"
    src.write_text("sequence,label\nACDEFGHIK,0\nACDEFG,1\nACDXFG,0\nACDEFGH,1\n")     # X is outside the alphabet
    out = tmp_path / "out.csv"
    p = _run(["--input", str(src), "--output", str(out)])
    assert p.returncode == 1
    assert "Alignment time" not in p.stdout
    assert "row 4" in p.stderr, p.stderr      # file line 4: the header is line 1
    # settings the reference's GUI would refuse are refused before any work (exit code 2)
    p = _run(["--input", str(src), "--output", str(out), "--gap", "3"])
    assert p.returncode == 2 and "settings" in p.stderr
    # a missing input file
    p = _run(["--input", str(tmp_path / "nothing.csv"), "--output", str(out)])
    assert p.returncode == 1 and p.stderr.strip()


def test_bench_two_ranks_on_one_card(tmp_path):
    """The N > 1 path of bench.py (ranks started by the script itself, row ranges of equal cell count, maximum over
    the ranks, sums of pairs and cells) with both ranks on this box's one GPU and gloo in place of RCCL."""
    env = dict(os.environ, SA_BENCH_DEVICE="0", SA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-extra", "--cpu-sample", "0", "--workload", "peptides4k"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2
    assert len(out["kernel_ms_per_rank"]) == 2 and all(v > 0 for v in out["kernel_ms_per_rank"])
    assert out["config"]["pairs"] == 4000 * 3999 // 2
    # the whole schedule's cells, whatever the split
    import sequencealigner_amd as sa
    from sequencealigner_amd import synth
    seqs, _ = synth.generate(4000, 6, 50, 1)
    al = sa.Aligner(device=sa.DEVICE_NONE)
    al.upload(seqs)
    assert out["config"]["cells"] == al.cell_count(sa.TRIANGULAR, 0, al.rows(sa.TRIANGULAR))
    al.close()
    assert out["value"] > 0 and out["unit"] == "GCUPS"
