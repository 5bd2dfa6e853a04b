"""profiles/<round>_pmc.json from the counter summaries tools/profile_all.sh leaves under gpurun_out/.
usage: python tools/make_pmc_json.py <out.json> <round tag> <tag>:<workload>:<kernel name> ...
Copies the summaries it used to profiles/<tag>_*.txt / .csv."""
import glob, json, os, re, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, rnd = sys.argv[1], sys.argv[2]
result = {}
for spec in sys.argv[3:]:
    tag, workload, kernel = spec.split(":", 2)
    vals = {}
    for path in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc_*.txt"))):
        for line in open(path):
            if kernel[:30] not in line:
                continue
            m = re.search(r"(\S+)\s+sum=\S+ dispatches=\d+ per_dispatch=(\S+)", line)
            if m:
                vals[m.group(1)] = float(m.group(2))
        shutil.copy(path, os.path.join(root, "profiles", f"{rnd}_{os.path.basename(path)}"))
    ks = os.path.join(root, "gpurun_out", f"{tag}_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(root, "profiles", f"{rnd}_{tag}_kernel_stats.csv"))
    bl = os.path.join(root, "gpurun_out", f"{tag}_bench_line.json")
    if os.path.exists(bl):
        shutil.copy(bl, os.path.join(root, "profiles", f"{rnd}_{tag}_bench_line.json"))
    rec = {
        "source": f"profiles/{rnd}_{tag}_pmc_*.txt (rocprofv3 --pmc passes of `python3 bench.py --workload {workload} --steps 2 --warmup 1 "
                  "--no-extra --cpu-sample 0`, one pass per counter group, per-launch averages)",
        "valu_wave_insts": vals.get("SQ_INSTS_VALU"),
        "salu_wave_insts": vals.get("SQ_INSTS_SALU"),
        "lds_wave_insts": vals.get("SQ_INSTS_LDS"),
        # SQ_INSTS_LDS_*_BANDWIDTH count 64-byte units (8 per ds_read_b64 of a 64-lane wave)
        "lds_bytes": (vals.get("SQ_INSTS_LDS_LOAD_BANDWIDTH", 0) + vals.get("SQ_INSTS_LDS_STORE_BANDWIDTH", 0)) * 64 or None,
        "lds_bank_conflict_share": round(vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"], 4) if vals.get("SQ_LDS_IDX_ACTIVE") else None,
        # FETCH_SIZE in KiB, doubled: gfx950 tallies 128-byte read requests at 64 bytes; WRITE_SIZE in KiB as read
        "fetch_bytes": vals["FETCH_SIZE"] * 1024 * 2 if "FETCH_SIZE" in vals else None,
        "write_bytes": vals["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in vals else None,
        "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in vals else None,
        "wave_quad_cycles": vals.get("SQ_WAVE_CYCLES"),
        "wait_any_share": round(vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], 4) if vals.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in vals else None,
        "wait_issue_share": round(vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"], 4) if vals.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in vals else None,
    }
    result[f"{kernel}|{workload}"] = rec
json.dump(result, open(out_path, "w"), indent=1)
print(json.dumps(result, indent=1))
