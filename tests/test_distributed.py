"""The N > 1 path on CPU: two gloo ranks shard the schedule rows the way bench.py
does, with no data-path collective; the shards tile the pair space exactly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["SA_ROOT"])
import torch
import torch.distributed as dist
import sequencealigner_amd as sa
from sequencealigner_amd import synth
import bench

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seqs, _ = synth.generate(3000, 6, 50, seed=1)
al = sa.Aligner(device=sa.DEVICE_NONE)
al.upload(seqs)
r0, r1 = bench.shard_for_rank(al, sa.TRIANGULAR, rank, world)
pairs = al.pair_count(sa.TRIANGULAR, r0, r1)
cells = al.cell_count(sa.TRIANGULAR, r0, r1)
t = torch.tensor([pairs, cells, r0, r1], dtype=torch.int64)
gathered = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
tot = bench.reduce_sum(torch.tensor([pairs, cells], dtype=torch.float64))
mx = bench.reduce_max(float(rank + 1))
if rank == 0:
    rows = al.rows(sa.TRIANGULAR)
    bounds = [(int(g[2]), int(g[3])) for g in gathered]
    assert bounds[0][0] == 0 and bounds[-1][1] == rows
    assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert int(tot[0]) == 3000 * 2999 // 2 == sum(int(g[0]) for g in gathered)
    assert int(tot[1]) == al.cell_count(sa.TRIANGULAR, 0, rows)
    assert mx == world
    cells_each = [int(g[1]) for g in gathered]
    assert max(cells_each) < 1.05 * sum(cells_each) / world
    print("OK")
dist.destroy_process_group()
"""


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SA_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
         "127.0.0.1", "--master-port", "29617", str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
