#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command -> gpurun_out/prof_<tag>/ ; prints the top of the kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_bench.err
F=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/prof_${TAG}_kernel_stats.csv
head -14 $F
