#!/bin/bash
# A/B of library variants on ONE box (box-to-box spread of the bench is ~2 %: only same-box comparisons mean anything).
#   bash scripts/ab_bench.sh _ab/base.so _ab/variant.so ...      (each variant is benched ROUNDS times, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    MBPO_HIP_LIB=$R/$lib python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],4), round(d['value']/1e6,3), round(d['roofline']['avg_launch_us'],2))" || exit 1
  done
done
