#!/bin/bash
# SQ counters of the bench command (no hipGraph, so every launch is a dispatch the profiler sees): where the waves' cycles go
# (parked at s_waitcnt / barrier, issue-stalled, issuing) and how busy the MFMA pipes are.  One pass, 8 SQ slots.
# Output: gpurun_out/pmc_sq/.../*counter_collection.csv  ->  python scripts/make_pmc_sq.py gpurun_out/pmc_sq profiles/r02_pmc_sq.json
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_sq
rm -rf $d
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $d -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-graph > $R/gpurun_out/pmc_sq.log 2>&1
echo "SQ pass done: $(find $d -name '*counter_collection.csv' | head -1)"
