#!/usr/bin/env python3
"""Headline benchmark: all-vs-all Needleman-Wunsch (score + traceback) of the
synthetic 50 000-peptide set (len 6-50, seed 1; BASELINE.json configs[2]) on N MI355X.

A step is one pass of the hot path over the whole pair space owned by the rank:
fill + traceback kernels, compact result records left in HBM, inputs resident
in HBM before the timed region.  With N > 1 the triangular pair space is split
into N contiguous row ranges of equal cell count, one per rank, with no
collective on the data path (total work is fixed: strong scaling).

Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import sequencealigner_amd as sa  # noqa: E402
from sequencealigner_amd import synth  # noqa: E402

# INT32 VALU peak used for the roofline: 256 CUs x 64 lanes x 2.4 GHz.  Measured on
# the device with tools/valu_peak.hip: 35-40 T lane-instructions/s for max3 / compare
# / alignbit / packed ops, 55-64 T for plain add / sub / logic (profiles/r01_valu_issue_rates.txt).
PEAK_TIOPS = 256 * 64 * 2.4e9 / 1e12
OPS_PER_CELL = 5          # 3 adds + 2 max of the reference recurrence (include/seqalign.h:55-58)
# Memory-side bytes per launch of the dominant kernel, from separate rocprofv3 --pmc passes of this
# same command (FETCH_SIZE doubled as the gfx950 counter requires, WRITE_SIZE as read); the passes
# and their output are kept under profiles/.  Reported as roofline.traffic when the file is there.
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")


def measured_traffic(kernel, workload):
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
        if rec.get("kernel") == kernel and rec.get("workload") == workload:
            return rec.get("bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None

WORKLOADS = {
    # name: (n, lo, hi, seed, description)
    "peptides50k": (50000, 6, 50, 1, "synthetic 50k peptides len 6-50 seed 1, all-vs-all (BASELINE.json configs[2])"),
    "peptides20k": (20000, 6, 50, 1, "synthetic 20k peptides len 6-50 seed 1, all-vs-all"),
    "peptides4k": (4000, 6, 50, 1, "synthetic 4k peptides len 6-50 seed 1, all-vs-all"),
    "proteins10k": (10000, 300, 1000, 2, "synthetic 10k proteins len 300-1000 seed 2, all-vs-all (BASELINE.json configs[4])"),
    "proteins2k": (2000, 300, 1000, 2, "first 2000 of the synthetic proteins len 300-1000 seed 2, all-vs-all"),
}


def shard_for_rank(al, schedule, rank, world):
    """Contiguous row range of the schedule owned by `rank`: equal cells per rank."""
    b = al.shard_rows(schedule, world)
    return b[rank], b[rank + 1]


def reduce_max(x):
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([x], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(x)


def reduce_sum(t):
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu()
    return t


def barrier_sync():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def cpu_baseline(seqs, labels, sample_n):
    """The CPU restatement (oracle/nw_oracle.c, the reference's multithreaded path
    re-stated) timed on the host cores of this box on a bounded sample of the same
    workload: the first `sample_n` sequences, all-vs-all, without writing."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    threads = min(16, os.cpu_count() or 1)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sample.csv")
        synth.write_csv(path, seqs[:sample_n], labels[:sample_n])
        st = oracle_lib.run_file(path, None, 1, threads=threads)
    return {
        "value": round(st.cells / st.seconds / 1e9, 3),
        "unit": "GCUPS",
        "pairs_per_s": round(st.pairs / st.seconds, 1),
        "cores": threads,
        "kind": "port",
        "seconds": round(st.seconds, 3),
        "sample": f"first {sample_n} of the {len(seqs)} sequences, all-vs-all ({st.pairs} pairs, {st.cells} cells), "
                  f"score + traceback, no CSV write, {threads} threads",
    }


def side_measurement(name, device):
    """A second workload measured after the timed region (not part of `value`): the
    long-sequence kernel on a prefix of the protein set of BASELINE.json configs[4]."""
    n, lo, hi, seed, desc = WORKLOADS[name]
    seqs, _ = synth.generate(n, lo, hi, seed)
    al = sa.Aligner(device=device, max_seq_len=hi)
    al.upload(seqs)
    rows = al.rows(sa.TRIANGULAR)
    cells = al.cell_count(sa.TRIANGULAR, 0, rows)
    pairs = al.pair_count(sa.TRIANGULAR, 0, rows)
    al.prepare_resident(sa.TRIANGULAR, 0, rows)
    al.run_resident()
    ms = [al.run_resident() for _ in range(2)]
    al.close()
    k_ms = sum(ms) / len(ms)
    return {"workload": desc, "pairs": pairs, "cells": cells, "kernel_ms": round(k_ms, 3),
            "gcups": round(cells / k_ms / 1e6, 1), "pairs_per_s": round(pairs / k_ms * 1e3, 1),
            "roofline_frac": round(cells * OPS_PER_CELL / (k_ms * 1e-3) / 1e12 / PEAK_TIOPS, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="peptides50k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-extra", action="store_true", help="skip the protein side measurement")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="sequences in the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    n, lo, hi, seed, desc = WORKLOADS[args.workload]
    seqs, labels = synth.generate(n, lo, hi, seed)
    al = sa.Aligner(device=local_rank, max_seq_len=max(64, hi))   # raises if the HIP library or the device is missing
    al.upload(seqs)
    r0, r1 = shard_for_rank(al, sa.TRIANGULAR, rank, world)
    my_pairs = al.pair_count(sa.TRIANGULAR, r0, r1)
    my_cells = al.cell_count(sa.TRIANGULAR, r0, r1)
    al.prepare_resident(sa.TRIANGULAR, r0, r1)

    for _ in range(args.warmup):
        al.run_resident()
    barrier_sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(al.run_resident())     # launches on the library's stream and waits for it
    barrier_sync()
    elapsed = reduce_max(time.perf_counter() - t0)
    totals = reduce_sum(torch.tensor([float(my_pairs), float(my_cells)], dtype=torch.float64))
    total_pairs, total_cells = float(totals[0]), float(totals[1])

    if rank == 0:
        k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        if hi > 64:
            kernel = "sa_long_pair_kernel<4>"
        else:
            kernel = "sa_peptide_pair_kernel<13, 4>" if hi <= 52 else "sa_peptide_pair_kernel<16, 4>"
        achieved = my_cells * OPS_PER_CELL / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic = measured_traffic(kernel, args.workload) if world == 1 and k_ms > 0 else None
        out = {
            "metric": "GCUPS, all-vs-all Needleman-Wunsch score+traceback (aligned pairs/s alongside)",
            "value": round(total_cells * args.steps / elapsed / 1e9, 2),
            "unit": "GCUPS",
            "pairs_per_s": round(total_pairs * args.steps / elapsed, 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "sequences": n,
                "pairs": int(total_pairs),
                "cells": int(total_cells),
                "schedule": "triangular, contiguous row ranges of equal cell count per GPU",
                "results": f"{al.stride}-byte record per pair (score, matches, gaps, traceback) left in HBM",
            },
            "roofline": {
                "bound": "valu_int32",
                "achieved": round(achieved, 3),
                "peak": round(PEAK_TIOPS, 2),
                "unit": "TIOP/s",
                "frac": round(achieved / PEAK_TIOPS, 4),
                "traffic": traffic,
                "memory_side": None if traffic is None else {
                    "achieved": round(traffic / (k_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(traffic / (k_ms * 1e-3) / 8e12, 4)},
                "kernel": kernel,
                "kernel_ms": round(k_ms, 3),
                "cells_per_launch": int(my_cells),
                "ops_per_cell": OPS_PER_CELL,
                "note": "INT32 VALU issue roofline: algorithmic ops = 5 per DP cell (3 adds + 2 max of the reference "
                        "recurrence); the kernels pack two cells per lane (exact 16-bit arithmetic, 3.9 instructions "
                        "issued per cell all told) and are still priced against the INT32 line, so frac can pass 1; "
                        "traffic = memory-side bytes per launch from rocprofv3 PMC passes (profiles/), null when not "
                        "measured; memory_side prices that traffic (direction rows written by the fill and read back "
                        "by the traceback, plus the records) against the 8 TB/s HBM line",
            },
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(seqs, labels, min(args.cpu_sample, n) if hi <= 64 else min(400, n))
        if world == 1 and args.workload == "peptides50k" and not args.no_extra:
            out["extra"] = {"proteins2k": side_measurement("proteins2k", local_rank)}
        print(json.dumps(out), flush=True)
    al.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
