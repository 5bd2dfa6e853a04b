#!/bin/bash
# A/B timing of library builds on one box: tools/ab.sh <workload> <libA.so> <libB.so> ...   (two rounds, alternating)
workload=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    SA_HIP_LIBRARY=$PWD/$lib timeout -k 10 120 python bench.py --workload $workload --steps 5 --warmup 1 --no-extra --cpu-sample 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['kernel_ms'], 'ms', d['value'], 'GCUPS')" || exit 1
  done
done
