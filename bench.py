#!/usr/bin/env python3
"""Headline benchmark: all-vs-all Needleman-Wunsch (score + traceback) of the
synthetic 50 000-peptide set (len 6-50, seed 1; BASELINE.json configs[2]) on N MI355X.

A step is one pass of the hot path over the whole pair space owned by the rank:
fill + traceback kernels, compact result records left in HBM, inputs resident
in HBM before the timed region.  With N > 1 the triangular pair space is split
into N contiguous row ranges of equal cell count, one per rank, with no
collective on the data path (total work is fixed: strong scaling).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself
(before anything touches a GPU) and passes their output through.

Rank 0 prints one JSON line.  Besides the contract's fields it carries
  roofline      the dominant kernel against the INT32 VALU line (5 ops per DP cell), plus the
                issue rate (VALU instructions actually issued), the LDS rate and the memory-side
                traffic, from rocprofv3 counter passes of this command kept under profiles/
  cpu_baseline  the reference's own multithreaded program (compiled from its sources into
                oracle/_ref) on the mega data set, and this repository's CPU restatement on a
                bounded sample of the bench workload, both on this box's host cores
  extra         the same workload on the 32-bit kernel, through PCIe to pinned host memory
                (the reference's "without write"), the 4 001 280-row data set end to end into
                a CSV file (its "with write") and resident on the ADJACENT kernel, the reference's
                own 1 042-sequence data set all-vs-all end to end, and the protein workloads
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# INT32 VALU peak used for the roofline: 256 CUs x 64 lanes x 2.4 GHz.  Measured on
# the device with tools/valu_peak.hip: 35-40 T lane-instructions/s for max3 / compare
# / alignbit / packed ops, 55-64 T for plain add / sub / logic (profiles/r01_valu_issue_rates.txt).
PEAK_TIOPS = 256 * 64 * 2.4e9 / 1e12
OPS_PER_CELL = 5          # 3 adds + 2 max of the reference recurrence (include/seqalign.h:55-58)
LDS_PEAK_TBS = 150.0      # all CUs streaming ds_read_b64 / b128 at 2.4 GHz (256 B per clock and CU)
HBM_PEAK_GBS = 8000.0
# Per-launch counter totals of the kernels, from separate rocprofv3 --pmc passes of this same command
# (tools/pmc_pass.sh; FETCH_SIZE doubled as gfx950 requires, WRITE_SIZE as read).  The passes' summaries
# are kept under profiles/; this file is refreshed whenever a kernel changes.
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc.json")
MIX_PEAK_FILE = "profiles/r02_valu_cell_block_rates.txt"      # tools/valu_peak.hip: the packed column-pair block back to back
# nominal: 256 CUs x 256 B per clock x 2.4 GHz (ds_read_b64; ds_read2_b64 and b128 run at half of it, measured on the
# values-only row of tools/row_bench.hip, profiles/r03_row_bench.txt)
LDS_PEAK_FILE = "nominal 256 CUs x 256 B/clk x 2.4 GHz; ds_read_b64 rows with and without their LDS reads in profiles/r03_row_bench.txt"
AVPDB_TRIANGULAR_SHA256 = "3eb3798aa4d2adff6aabe7bf69f06ef1425d2c25005b456f2a6440eff47e1e08"   # the reference's own output
AVPDB_CELLS = 252634121
MEGA_SHA256 = "667f3524a9ba31243de55c075165b336010ffeaf76600b39540b54f5dd085f18"   # the reference's own output, 500 832 027 bytes
MEGA_CELLS = 2214981084

WORKLOADS = {
    # name: (n, lo, hi, seed, description)
    "peptides50k": (50000, 6, 50, 1, "synthetic 50k peptides len 6-50 seed 1, all-vs-all (BASELINE.json configs[2])"),
    "peptides_g7": (50000, 25, 28, 1, "synthetic 50k peptides len 25-28 seed 1, all-vs-all (experiment: one row shape)"),
    "peptides20k": (20000, 6, 50, 1, "synthetic 20k peptides len 6-50 seed 1, all-vs-all"),
    "peptides4k": (4000, 6, 50, 1, "synthetic 4k peptides len 6-50 seed 1, all-vs-all"),
    "proteins10k": (10000, 300, 1000, 2, "synthetic 10k proteins len 300-1000 seed 2, all-vs-all (BASELINE.json configs[4])"),
    "proteins2k": (2000, 300, 1000, 2, "first 2000 of the synthetic proteins len 300-1000 seed 2, all-vs-all"),
    "domains16k": (16000, 65, 128, 3, "synthetic 16k sequences len 65-128 seed 3, all-vs-all (between the two regimes)"),
    # the reference's enlarged data set (BASELINE.json configs[3]): no generator, read from the golden avpdb.csv
    "mega": (4001280, 2, 46, None, "mega data set: avpdb.csv forwards + backwards x 1920, 4 001 279 adjacent pairs"),
    # few long pairs: the workgroup-per-pair kernel's case
    "adjacent_65x4096": (65, 4096, 4096, 5, "65 synthetic sequences of 4096 residues seed 5, 64 adjacent pairs"),
}
ADJACENT_WORKLOADS = ("mega", "adjacent_65x4096")


def workload_sequences(sa, synth, name):
    """(sequences, labels, schedule) of a workload."""
    n, lo, hi, seed, _ = WORKLOADS[name]
    if name == "mega":
        with open(os.path.join(ROOT, "tests", "golden", "avpdb.csv"), "rb") as f:
            rows = [line.split(b",") for line in f.read().split(b"\n")[1:] if line]
        seqs = [r[0] for r in rows]
        labels = [int(r[1]) for r in rows]
        return (seqs + seqs[::-1]) * 1920, (labels + labels[::-1]) * 1920, sa.ADJACENT
    seqs, labels = synth.generate(n, lo, hi, seed)
    return seqs, labels, sa.ADJACENT if name in ADJACENT_WORKLOADS else sa.TRIANGULAR


def visible_gpus():
    """GPUs of this machine counted from the kernel driver's topology files (nodes with shader engines), without
    loading torch or any HIP library: the process that starts the ranks must not have initialised a GPU.
    Without the driver's topology directory there is no GPU this library could use: 0."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return 0
    count = 0
    for node in nodes:
        try:
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            count += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            count = min(count, len([d for d in os.environ[var].split(",") if d.strip()]))
    return count


def spawn_ranks(args):
    """--gpus N outside a torchrun environment: start the N ranks as a fresh child job.  This process never
    touches a GPU: it imports neither torch nor the library, and counts devices from sysfs."""
    have = visible_gpus()
    if have < args.gpus and "SA_BENCH_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {args.gpus} asked for, {have} visible", file=sys.stderr)
        return 2
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def pmc_record(kernel, workload):
    try:
        with open(PMC_FILE) as f:
            return json.load(f).get(f"{kernel}|{workload}")
    except (OSError, ValueError):
        return None


def shard_for_rank(al, schedule, rank, world):
    """Contiguous row range of the schedule owned by `rank`: equal cells per rank."""
    b = al.shard_rows(schedule, world)
    return b[rank], b[rank + 1]


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _on_device(t):
    dist = _dist()
    return t.cuda() if dist is not None and dist.get_backend() == "nccl" else t


def reduce_max(x):
    """Maximum of a float over the ranks (the value itself without a process group)."""
    import torch
    dist = _dist()
    if dist is None:
        return float(x)
    t = _on_device(torch.tensor([x], dtype=torch.float64))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(t):
    """Element-wise sum of a tensor over the ranks, returned on the host."""
    dist = _dist()
    if dist is None:
        return t
    t = _on_device(t.clone())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu()


def per_rank(x):
    """The ranks' values of a float, in rank order."""
    import torch
    dist = _dist()
    if dist is None:
        return [float(x)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64)
    t[dist.get_rank()] = x
    return [float(v) for v in reduce_sum(t)]


def scratch_dir():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(sa, synth, seqs, labels, sample_n, mega_path):
    """Host-core baselines of this box, outside the timed region.
    kind "reference": oracle/_ref/ref_main_mt_nowrite, the reference's src/main.c + include/thread.h pool compiled
    from its own sources with its release flags, MODE_MULTITHREAD 1, MODE_WRITE 0, on the mega data set (its own
    headline case, README.md:165-170); it picks min(online CPUs, 16) threads itself (include/thread.h:57-66).
    "port": oracle/nw_oracle.c (this repository's restatement) on the first `sample_n` sequences of the bench
    workload, all-vs-all, no write, at min(cores, 16) threads and at every core this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    cores = host_cores()
    out = {}
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_main_mt_nowrite")
    if mega_path and os.path.exists(ref_bin):
        best = None
        for _ in range(2):
            env = dict(os.environ, SA_REF_INPUT=mega_path, SA_REF_OUTPUT=os.devnull)
            p = subprocess.run([ref_bin], env=env, capture_output=True, text=True, timeout=300)
            m = re.search(r"Alignment time: ([0-9.]+) seconds", p.stdout)
            if p.returncode == 0 and m:
                t = float(m.group(1))
                best = t if best is None else min(best, t)
        if best:
            out = {"value": round(MEGA_CELLS / best / 1e9, 3), "unit": "GCUPS", "cores": min(cores, 16), "kind": "reference",
                   "pairs_per_s": round(4001279 / best, 1), "seconds": round(best, 4),
                   "sample": "the reference's multithreaded program (src/main.c, include/thread.h pool) built from its sources, "
                             "MODE_WRITE 0, on the 4 001 280-row mega data set (adjacent pairs, 4 001 279 pairs, "
                             f"{MEGA_CELLS} cells), best of 2 runs, min(online CPUs, 16) threads"}
    with tempfile.TemporaryDirectory(dir=scratch_dir()) as d:
        path = os.path.join(d, "sample.csv")
        synth.write_csv(path, seqs[:sample_n], labels[:sample_n])
        ports = []
        for threads in sorted({min(cores, 16), min(cores, 64)}):
            st = oracle_lib.run_file(path, None, 1, threads=threads)
            ports.append({"value": round(st.cells / st.seconds / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
                          "pairs_per_s": round(st.pairs / st.seconds, 1), "seconds": round(st.seconds, 3),
                          "sample": f"first {sample_n} of the {len(seqs)} sequences, all-vs-all ({st.pairs} pairs, {st.cells} cells), "
                                    f"score + traceback, no CSV write, {threads} threads"})
    # BASELINE.json configs[0]: the reference's own CPU-runnable case, avpdb.csv all-vs-all, on the restatement
    threads = min(cores, 16)
    st = oracle_lib.run_file(os.path.join(ROOT, "tests", "golden", "avpdb.csv"), None, 1, threads=threads)
    avpdb = {"value": round(st.cells / st.seconds / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
             "pairs_per_s": round(st.pairs / st.seconds, 1), "seconds": round(st.seconds, 4), "cells": st.cells,
             "sample": f"avpdb.csv all-vs-all ({st.pairs} pairs, {st.cells} cells), score + traceback, no CSV write, {threads} threads"}
    if out:
        out["port"] = ports
    else:
        out = dict(ports[0])
        out["port"] = ports[1:]
    out["avpdb_all_pairs"] = avpdb
    out["host_cores_available"] = cores
    return out


def resident_measurement(sa, synth, name, device, launches=2, pairing=True, checkpoint_rows=None):
    """A workload measured with the records left in HBM (not part of `value`)."""
    n, lo, hi, seed, desc = WORKLOADS[name]
    seqs, _, schedule = workload_sequences(sa, synth, name)
    al = sa.Aligner(device=device, max_seq_len=max(64, hi))
    al.set_pairing(pairing)
    if checkpoint_rows is not None:
        al.set_long_checkpoint_rows(checkpoint_rows)
    al.upload(seqs)
    rows = al.rows(schedule)
    cells = al.cell_count(schedule, 0, rows)
    pairs = al.pair_count(schedule, 0, rows)
    al.prepare_resident(schedule, 0, rows)
    al.run_resident()
    ms = [al.run_resident() for _ in range(launches)]
    kernel = al.last_kernel()
    grid = al.last_grid()
    al.close()
    k_ms = sum(ms) / len(ms)
    out = {"workload": desc, "kernel": kernel, "workgroups": grid[0], "workgroups_wanted": grid[1], "pairs": pairs, "cells": cells,
           "kernel_ms": round(k_ms, 3),
           "gcups": round(cells / k_ms / 1e6, 1), "pairs_per_s": round(pairs / k_ms * 1e3, 1),
           "roofline_frac": round(cells * OPS_PER_CELL / (k_ms * 1e-3) / 1e12 / PEAK_TIOPS, 4)}
    if "pair_kernel" in kernel or "ckpt" in kernel:
        out["roofline_frac_packed16"] = round(out["roofline_frac"] / 2, 4)
    rec = pmc_record(kernel, name)
    if rec:
        out["counters"] = counter_rates(rec, k_ms)
    return out


def mixed_lengths_measurement(sa, synth, device):
    """Peptides with a few long sequences among them: the set is a long-sequence one (kernel, record form and stride
    follow the longest sequence); the tiles whose sequences are all peptides run on the peptide kernel (on), or every
    tile on the long-sequence kernel (off)."""
    out = {}
    for key, (npep, extra, lo, hi) in {"20000_peptides_plus_20_of_80_120": (20000, 20, 80, 120),
                                       "8000_peptides_plus_20_of_300_1000": (8000, 20, 300, 1000)}.items():
        seqs, _ = synth.generate(npep, 6, 50, 3)
        more, _ = synth.generate(extra, lo, hi, 5)
        step = len(seqs) // extra
        for k, s in enumerate(more):
            seqs.insert(k * step + k, s)
        al = sa.Aligner(device=device, max_seq_len=hi)
        al.upload(seqs)
        rows = al.rows(sa.TRIANGULAR)
        res = {"pairs": al.pair_count(sa.TRIANGULAR, 0, rows), "cells": al.cell_count(sa.TRIANGULAR, 0, rows), "record_bytes": al.stride}
        for tiles, name in ((True, "peptide_tiles_on"), (False, "peptide_tiles_off")):
            al.set_peptide_tiles(tiles)
            al.prepare_resident(sa.TRIANGULAR, 0, rows)
            al.run_resident()
            ms = min(al.run_resident() for _ in range(2))
            res[name] = {"kernel": al.last_kernel(), "kernel_ms": round(ms, 3), "gcups": round(res["cells"] / ms / 1e6, 1)}
        al.close()
        out[key] = res
    return out


def few_pairs_measurement(sa, synth, device):
    """Submissions of few long pairs, where a pair per lane leaves the chip idle: one wave per pair, a workgroup of
    waves per pair (the library picks 1 to 16 waves from the pair count) and the pair-per-lane kernel on 64 adjacent
    pairs of 4096-residue sequences and on 150 proteins all-vs-all."""
    out = {}
    cases = {"adjacent_65x4096": (65, 4096, 4096, 5, sa.ADJACENT), "triangular_150_proteins": (150, 300, 1000, 2, sa.TRIANGULAR)}
    for name, (n, lo, hi, seed, schedule) in cases.items():
        seqs, _ = synth.generate(n, lo, hi, seed)
        al = sa.Aligner(device=device, max_seq_len=hi)
        al.upload(seqs)
        rows = al.rows(schedule)
        res = {"pairs": al.pair_count(schedule, 0, rows), "cells": al.cell_count(schedule, 0, rows)}
        for mode, waves, key in ((1, 1, "wave_per_pair"), (1, 0, "workgroup_per_pair"), (-1, 0, "pair_per_lane")):
            al.set_cooperative(mode)
            al.set_cooperative_waves(waves)
            al.prepare_resident(schedule, 0, rows)
            al.run_resident()
            ms = min(al.run_resident() for _ in range(3))
            res[key] = {"kernel": al.last_kernel(), "kernel_ms": round(ms, 3), "gcups": round(res["cells"] / ms / 1e6, 1)}
        al.set_cooperative(0)
        al.prepare_resident(schedule, 0, rows)
        al.run_resident()
        res["library_choice"] = al.last_kernel()
        al.close()
        out[name] = res
    return out


MIX_PEAK_TLANE = 38.49


def counter_rates(rec, k_ms):
    """Rates from the per-launch counter totals of profiles/ and the kernel time measured in this run."""
    sec = k_ms * 1e-3
    out = {"source": rec.get("source")}
    if rec.get("valu_wave_insts"):
        lane = rec["valu_wave_insts"] * 64
        out["issue"] = {"lane_insts_per_launch": lane, "achieved": round(lane / sec / 1e12, 2), "peak": round(PEAK_TIOPS, 2),
                        "unit": "T lane-instr/s", "frac": round(lane / sec / 1e12 / PEAK_TIOPS, 4),
                        # the packed kernels' column-pair block issued back to back with no memory access, four waves per
                        # SIMD (tools/valu_peak.hip, profiles/r02_valu_cell_block_rates.txt): the chip lowers its clock to
                        # 1.94 GHz under that stream, so this, not the 2.4 GHz line, is what the instruction mix can reach
                        "mix_peak": MIX_PEAK_TLANE, "mix_peak_from": MIX_PEAK_FILE,
                        "mix_frac": round(lane / sec / 1e12 / MIX_PEAK_TLANE, 4)}
    if rec.get("lds_bytes"):
        out["lds"] = {"bytes_per_launch": rec["lds_bytes"], "achieved": round(rec["lds_bytes"] / sec / 1e12, 2),
                      "peak": LDS_PEAK_TBS, "peak_from": LDS_PEAK_FILE, "unit": "TB/s", "frac": round(rec["lds_bytes"] / sec / 1e12 / LDS_PEAK_TBS, 4),
                      "bank_conflict_share": rec.get("lds_bank_conflict_share")}
    if rec.get("fetch_bytes") is not None and rec.get("write_bytes") is not None:
        t = rec["fetch_bytes"] + rec["write_bytes"]
        out["memory_side"] = {"bytes_per_launch": t, "achieved": round(t / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(t / sec / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def d2h_measurement(sa, csv_path, cells):
    """The bench workload with every record copied to pinned host memory (sa_submit / sa_next, two submissions in
    flight, nothing formatted or written): the reference's "without write" mode (MODE_WRITE 0, src/main.c:111-116)."""
    rep = sa.run_csv(csv_path, None, sa.TRIANGULAR)
    full = sa.run_csv(csv_path, None, sa.TRIANGULAR, compact_records=0)
    return {"what": "CSV ingest + upload + kernels + every record over PCIe to pinned host memory, no formatting, no write; records "
                    "packed on the device first (header + one byte per row of the pair's second sequence, 4-byte aligned)",
            "seconds": round(rep.seconds, 4), "gcups": round(cells / rep.seconds / 1e9, 1),
            "pairs_per_s": round(rep.pairs / rep.seconds, 1),
            "record_bytes": rep.record_bytes, "record_gb_per_s": round(rep.record_bytes / rep.seconds / 1e9, 2),
            "full_stride_records": {"seconds": round(full.seconds, 4), "record_bytes": full.record_bytes,
                                    "gcups": round(cells / full.seconds / 1e9, 1)},
            "seconds_ingest": round(rep.seconds_ingest, 4), "seconds_upload": round(rep.seconds_upload, 4),
            "seconds_gpu_wait": round(rep.seconds_gpu_wait, 4)}


def e2e_avpdb(sa, out_dir, device):
    """BASELINE.json configs[1]: the reference's own data set (1 042 sequences), all-vs-all, CSV in -> CSV out, checked
    against the sha256 of the reference's output.  A job of half a million pairs: set-up and the write are most of it."""
    src = os.path.join(ROOT, "tests", "golden", "avpdb.csv")
    dst = os.path.join(out_dir, "avpdb_triangular.csv")
    st = sa.default_settings()
    st.num_devices = 1
    st.devices[0] = device
    best = min((sa.run_csv(src, dst, sa.TRIANGULAR, settings=st) for _ in range(3)), key=lambda r: r.seconds)
    ok = sha256_file(dst) == AVPDB_TRIANGULAR_SHA256
    os.unlink(dst)
    nowrite = min((sa.run_csv(src, None, sa.TRIANGULAR, settings=st) for _ in range(3)), key=lambda r: r.seconds)
    return {"what": "avpdb.csv all-vs-all (542 361 pairs) CSV in -> CSV out, set-up excluded as in the reference's timing",
            "seconds": round(best.seconds, 4), "pairs_per_s": round(best.pairs / best.seconds, 1),
            "gcups": round(best.cells / best.seconds / 1e9, 2), "cells": best.cells, "bytes_written": best.bytes_written,
            "sha256_ok": ok, "without_write": {"seconds": round(nowrite.seconds, 4), "gcups": round(nowrite.cells / nowrite.seconds / 1e9, 2)}}


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def e2e_mega(sa, mega_path, out_dir, device):
    """BASELINE.json configs[3]: the 4 001 280-row data set, adjacent pairs, CSV in -> CSV out ("with write",
    README.md:165-170), output checked against the sha256 of the reference's own output.  One context, then three
    contexts allowed on this one card: the driver uses as many of them as the expected kernel time can feed (one per 30 ms),
    which for this job is one; "contexts_3_forced" makes it use all three, to show what further contexts cost a job whose
    time goes to the host (clones of the sequences, pinned buffers, three writers on one file)."""
    out = {"what": "mega data set (4 001 279 adjacent pairs) CSV in -> 500 832 027-byte CSV out, set-up excluded as in the "
                   "reference's timing (README.md:178)"}
    for ndev, forced in ((1, 0), (3, 0), (3, 1)):
        st = sa.default_settings()
        st.num_devices = ndev
        st.force_devices = forced
        for k in range(ndev):
            st.devices[k] = device
        dst = os.path.join(out_dir, f"mega_out_{ndev}.csv")
        best = None
        for _ in range(3):
            rep = sa.run_csv(mega_path, dst, sa.ADJACENT, settings=st)
            if best is None or rep.seconds < best.seconds:
                best = rep
        ok = sha256_file(dst) == MEGA_SHA256
        os.unlink(dst)
        out[f"contexts_{ndev}" + ("_forced" if forced else "")] = {
            "devices_used": best.devices_used, "seconds": round(best.seconds, 4), "pairs_per_s": round(best.pairs / best.seconds, 1),
            "gcups": round(best.cells / best.seconds / 1e9, 2), "bytes_written": best.bytes_written, "sha256_ok": ok,
            "seconds_ingest": round(best.seconds_ingest, 4), "seconds_upload": round(best.seconds_upload, 4),
            "device_wait": [round(best.device_wait[k], 4) for k in range(ndev)],
            "device_submit": [round(best.device_submit[k], 4) for k in range(ndev)],
            "device_write": [round(best.device_write[k], 4) for k in range(ndev)]}
    out["output_dir"] = out_dir
    # The same run with the output in the default temporary directory (a disk file system's page cache where out_dir is
    # a tmpfs: the two take a 500 MB write at different rates, and the write is what bounds this run)
    alt_dir = tempfile.gettempdir()
    try:
        if os.path.realpath(alt_dir) != os.path.realpath(out_dir) and shutil.disk_usage(alt_dir).free > (4 << 30):
            st = sa.default_settings()
            st.num_devices = 1
            st.devices[0] = device
            dst = os.path.join(alt_dir, f"sa_mega_out_{os.getpid()}.csv")
            try:
                rep = min((sa.run_csv(mega_path, dst, sa.ADJACENT, settings=st) for _ in range(2)), key=lambda r: r.seconds)
                ok = sha256_file(dst) == MEGA_SHA256
            finally:
                if os.path.exists(dst):
                    os.unlink(dst)
            out["contexts_1_other_dir"] = {"output_dir": alt_dir, "seconds": round(rep.seconds, 4),
                                           "pairs_per_s": round(rep.pairs / rep.seconds, 1), "sha256_ok": ok,
                                           "device_write": [round(rep.device_write[0], 4)]}
    except OSError:
        pass
    # MODE_WRITE 0 (the reference's "without write"): what is left is the host pipeline itself and, per further context
    # that is forced into use, its set-up (clone of the sequences, pinned and device buffers)
    for ndev, forced in ((1, 0), (3, 0), (3, 1)):
        st = sa.default_settings()
        st.num_devices = ndev
        st.force_devices = forced
        for k in range(ndev):
            st.devices[k] = device
        rep = min((sa.run_csv(mega_path, None, sa.ADJACENT, settings=st) for _ in range(5)), key=lambda r: r.seconds)
        key = "without_write" if ndev == 1 else "without_write_contexts_3" + ("_forced" if forced else "")
        out[key] = {"devices_used": rep.devices_used, "seconds": round(rep.seconds, 4),
                    "pairs_per_s": round(rep.pairs / rep.seconds, 1), "gcups": round(rep.cells / rep.seconds / 1e9, 2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="peptides50k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (32-bit kernel, PCIe, CSV end to end, proteins)")
    ap.add_argument("--cpu-sample", type=int, default=8000, help="sequences in the CPU restatement's sample (0 = no CPU baseline)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    import sequencealigner_amd as sa
    from sequencealigner_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    # Rehearsal knobs for a box with fewer GPUs than ranks (the N > 1 code path on one card): SA_BENCH_DEVICE pins every
    # rank to one device, SA_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one GPU.
    if "SA_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SA_BENCH_DEVICE"])
    if world > 1:
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("SA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            # the communicator is bound to this rank's GPU at once: barriers need not guess a device
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier_sync():
        if world > 1:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    n, lo, hi, seed, desc = WORKLOADS[args.workload]
    seqs, labels, schedule = workload_sequences(sa, synth, args.workload)
    al = sa.Aligner(device=local_rank, max_seq_len=max(64, hi))   # raises if the HIP library or the device is missing
    al.upload(seqs)
    r0, r1 = shard_for_rank(al, schedule, rank, world)
    my_pairs = al.pair_count(schedule, r0, r1)
    my_cells = al.cell_count(schedule, r0, r1)
    al.prepare_resident(schedule, r0, r1)

    for _ in range(args.warmup):
        al.run_resident()
    barrier_sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(al.run_resident())     # launches on the library's stream and waits for it (HIP events around the kernel)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
    kernel = al.last_kernel()
    elapsed = reduce_max(elapsed)
    totals = reduce_sum(torch.tensor([float(my_pairs), float(my_cells)], dtype=torch.float64))
    total_pairs, total_cells = float(totals[0]), float(totals[1])
    per_rank_ms = [round(v, 3) for v in per_rank(k_ms)]

    if rank == 0:
        achieved = my_cells * OPS_PER_CELL / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        rec = pmc_record(kernel, args.workload) if world == 1 and k_ms > 0 else None
        rates = counter_rates(rec, k_ms) if rec else {}
        mem = rates.get("memory_side")
        packed = "pair_kernel" in kernel or "ckpt" in kernel
        out = {
            "metric": "GCUPS, all-vs-all Needleman-Wunsch score+traceback (aligned pairs/s alongside)",
            "value": round(total_cells * args.steps / elapsed / 1e9, 2),
            "unit": "GCUPS",
            "pairs_per_s": round(total_pairs * args.steps / elapsed, 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "kernel_ms_per_rank": per_rank_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int16" if packed else "int32",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "sequences": n,
                "pairs": int(total_pairs),
                "cells": int(total_cells),
                "schedule": ("triangular" if schedule == sa.TRIANGULAR else "adjacent") + ", contiguous row ranges of equal cell count per GPU",
                "results": f"{al.stride}-byte record per pair (score, matches, gaps, traceback) left in HBM",
            },
            "notes": [
                "value is the kernels' throughput with the records left in HBM; what a caller receives over PCIe is extra.d2h",
                "BASELINE.json configs[3] (extra.e2e_mega: 4 001 279 adjacent pairs, CSV in -> CSV out) is host-bound: its kernel "
                "takes about 2 ms (extra.adjacent_kernel) of about 160 ms; ingest, upload and the 500 MB write are the rest, so "
                "further GPUs do not make it faster and the CSV driver gives such a job one context",
                "N > 1: every rank aligns its own row range of the triangle, no collective on the data path; two ranks have "
                "been run on one card (tests/test_gpu_cli.py), no 2-, 4- or 8-GPU figure has been measured by this repository",
            ],
            "roofline": {
                "bound": "valu",
                "bound_detail": "vector ALU issue: neither HBM nor MFMA bounds this path.  The packed kernels' recurrence is one v_add_u32 "
                                "and one v_pk_maximum3_f16 on 16-bit integer patterns per two cells (plus the tag fields); the 32-bit "
                                "kernels use v_add_u32 / v_max3_i32.  peak = the plain INT32 lane rate; a packed instruction does two "
                                "cells, so frac_packed16 (= achieved / (2 x peak)) is the fraction to read for the packed kernels",
                "achieved": round(achieved, 3),
                "peak": round(PEAK_TIOPS, 2),
                "unit": "TIOP/s",
                "frac": round(achieved / PEAK_TIOPS, 4),
                "frac_packed16": round(achieved / PEAK_TIOPS / 2, 4) if packed else None,
                "traffic": mem["bytes_per_launch"] if mem else None,
                "issue_frac": rates.get("issue", {}).get("frac"),
                "issue": rates.get("issue"),
                "lds": rates.get("lds"),
                "memory_side": mem,
                "counters_from": rates.get("source"),
                "kernel": kernel,
                "kernel_ms": round(k_ms, 3),
                "cells_per_launch": int(my_cells),
                "ops_per_cell": OPS_PER_CELL,
                "note": "frac prices the reference recurrence's 5 integer ops per DP cell (3 adds + 2 max) against the INT32 VALU line "
                        "(256 CUs x 64 lanes x 2.4 GHz); the packed kernels hold two 16-bit cells per lane (exact), so frac can pass 1 "
                        "-- issue_frac is the measure of how close the kernel sits to the hardware: VALU lane-instructions actually "
                        "issued per second over the same line.  kernel_ms is measured live (HIP events on the launch stream); the "
                        "instruction, LDS and memory-side byte counts per launch are constants of the kernel and workload taken from "
                        "rocprofv3 counter passes of this command (file named in counters_from, null when there is none for this "
                        "kernel); memory_side = L2-fabric bytes (direction rows written by the fill and read back by the traceback, "
                        "plus the records; Infinity-Cache hits included) against the 8 TB/s HBM line",
            },
        }
        if world == 1 and not args.no_extra:
            extra = {}
            tmp = tempfile.TemporaryDirectory(dir=scratch_dir())
            mega_path = None
            try:
                if args.workload == "peptides50k":
                    al.close()
                    extra["int32_kernel"] = resident_measurement(sa, synth, args.workload, local_rank, pairing=False)
                    csv_path = os.path.join(tmp.name, "peptides50k.csv")
                    synth.write_csv(csv_path, seqs, labels)
                    extra["d2h"] = d2h_measurement(sa, csv_path, int(total_cells))
                    mega_path = os.path.join(tmp.name, "mega.csv")
                    synth.make_mega(os.path.join(ROOT, "tests", "golden", "avpdb.csv"), mega_path)
                    extra["e2e_mega"] = e2e_mega(sa, mega_path, tmp.name, local_rank)
                    extra["adjacent_kernel"] = resident_measurement(sa, synth, "mega", local_rank, launches=5)
                    extra["e2e_avpdb_triangular"] = e2e_avpdb(sa, tmp.name, local_rank)
                    extra["few_long_pairs"] = few_pairs_measurement(sa, synth, local_rank)
                    extra["proteins2k"] = resident_measurement(sa, synth, "proteins2k", local_rank)
                    extra["proteins2k_direction_rows_kernel"] = resident_measurement(sa, synth, "proteins2k", local_rank, checkpoint_rows=0)
                    extra["proteins10k"] = resident_measurement(sa, synth, "proteins10k", local_rank, launches=1)
                    # between the two regimes: sequences a little longer than the peptide kernel's 64 columns
                    extra["domains16k"] = resident_measurement(sa, synth, "domains16k", local_rank, launches=1)
                    extra["mixed_lengths"] = mixed_lengths_measurement(sa, synth, local_rank)
                out["extra"] = extra
                if args.cpu_sample > 0:
                    out["cpu_baseline"] = cpu_baseline(sa, synth, seqs, labels, min(args.cpu_sample, n) if hi <= 64 else min(300, n),
                                                       mega_path)
            finally:
                tmp.cleanup()
        elif world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(sa, synth, seqs, labels, min(args.cpu_sample, n) if hi <= 64 else min(300, n), None)
        print(json.dumps(out), flush=True)
    al.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
