#!/bin/bash
# HBM traffic per launch: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only) of the bench command
# without hipGraph (so every launch is a dispatch the profiler sees).  Output: gpurun_out/pmc_{fetch,write}/.../*counter_collection.csv
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$(echo $c | tr A-Z a-z)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
  echo "$c done: $(find $d -name '*counter_collection.csv' | head -1)"
done
