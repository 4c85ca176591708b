#!/bin/bash
# End-of-round evidence in one GPU call: full GPU test suite, bench line, rocprofv3 kernel-trace stats of the same bench command,
# per-kernel rooflines.  Everything lands under gpurun_out/ (copy what is to be kept into profiles/).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/tests_gpu.log 2>&1 && tail -2 gpurun_out/tests_gpu.log
timeout -k 10 400 python bench.py --steps 50 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err && cat gpurun_out/bench.json
timeout -k 10 500 python scripts/bench_kernels.py --out gpurun_out/kernel_rooflines.json > /dev/null 2> gpurun_out/kernel_rooflines.err && tail -5 gpurun_out/kernel_rooflines.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -3
