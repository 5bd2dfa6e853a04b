#!/bin/bash
# One rocprofv3 --pmc pass of the bench command per counter group; summaries under gpurun_out/<tag>_<n>.txt.
# usage: tools/pmc_pass.sh <tag> <workload> "<counters of pass 1>" ["<counters of pass 2>" ...]
set -u
tag=$1; workload=$2; shift 2
cd /tmp && export TMPDIR=/tmp
n=0
for group in "$@"; do
  n=$((n+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$n
  rm -rf "$out"
  timeout -s KILL 150 rocprofv3 --pmc $group --output-format csv -d "$out" -- python3 $GRAFT_REPO_ROOT/bench.py --workload $workload --steps 2 --warmup 1 --no-extra --cpu-sample 0 > "$out.log" 2>&1 || { echo "pass $n failed"; tail -n 5 "$out.log"; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$out" > "$out.txt"
  grep -v -E "pack_slots|format" "$out.txt"
  rm -rf "$out"
done
