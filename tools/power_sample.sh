#!/bin/bash
# Samples socket power and shader clock (rocm-smi) while the bench's dominant kernel runs back to back.
# usage: tools/power_sample.sh <workload> <steps>   -> gpurun_out/power_<workload>.txt
workload=${1:-peptides50k}; steps=${2:-60}
out=gpurun_out/power_$workload.txt
mkdir -p gpurun_out
echo "idle:" > $out
rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|W\)" >> $out
timeout -k 10 200 python bench.py --workload $workload --steps $steps --warmup 1 --no-extra --cpu-sample 0 > gpurun_out/power_bench_$workload.json 2>/dev/null &
pid=$!
sleep 6     # import, data set, upload
echo "under load:" >> $out
for k in 1 2 3 4 5 6 7 8; do
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|W\)" >> $out
done
wait $pid
echo "bench exit $?" >> $out
python -c "import json; d=json.loads(open('gpurun_out/power_bench_$workload.json').read().strip().splitlines()[-1]); print('kernel_ms', d['roofline']['kernel_ms'])" >> $out
cat $out
