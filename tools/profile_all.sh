#!/bin/bash
# Kernel trace and counter passes of the bench command for one workload; summaries under gpurun_out/<tag>_*.
# usage: tools/profile_all.sh <tag> <workload>
set -u
tag=$1; workload=$2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
kt=$root/gpurun_out/${tag}_kt
rm -rf "$kt"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$kt" -- python3 $root/bench.py --workload $workload --steps 5 --warmup 1 --no-extra --cpu-sample 0 > "$kt.log" 2>&1 || { echo "kernel trace failed"; tail -n 5 "$kt.log"; exit 1; }
find "$kt" -name "*kernel_stats.csv" -exec cp {} $root/gpurun_out/${tag}_kernel_stats.csv \;
grep -h "^{" "$kt.log" > $root/gpurun_out/${tag}_bench_line.json
rm -rf "$kt"
cat $root/gpurun_out/${tag}_kernel_stats.csv
cd $root && tools/pmc_pass.sh ${tag}_pmc $workload \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"
