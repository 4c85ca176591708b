#!/bin/bash
# Second evidence call of a round (after scripts/collect_profiles.sh): PMC traffic + SQ passes of the bench command, kernel stats and
# SQ passes of the PPO (BASELINE config 3) and BPTT (config 5) steps issued eagerly, and the in-kernel s_memtime timelines.
# Everything lands under gpurun_out/; scripts/refresh_profiles.sh <tag> copies the summaries into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04}
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"
mkdir -p $R/gpurun_out
bash $R/scripts/collect_pmc.sh
bash $R/scripts/collect_pmc_sq.sh
cd /tmp && export TMPDIR=/tmp
for w in ppo_c3 bptt_c5; do
  arg=""; [ $w = ppo_c3 ] && arg=40
  rm -rf $R/gpurun_out/${w}_prof $R/gpurun_out/${w}_sq
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${w}_prof -- python3 $R/scripts/${w}_profile.py $arg > $R/gpurun_out/${w}_prof.log 2>&1
  cp $(find $R/gpurun_out/${w}_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${T}_${w}_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/${w}_sq -- python3 $R/scripts/${w}_profile.py $arg > $R/gpurun_out/${w}_sq.log 2>&1
  echo "$w done"
done
cd $R
timeout -k 10 200 python scripts/lean_dev.py > gpurun_out/${T}_sac_lean_stamps.txt 2>&1
timeout -k 10 200 python scripts/ppo_lean_dev.py > gpurun_out/${T}_ppo_lean_stamps.txt 2>&1
timeout -k 10 200 python scripts/step_flavours.py > gpurun_out/${T}_step_flavours.txt 2>&1
timeout -k 10 200 python scripts/sac_phase_stamps.py 128,128,128 > gpurun_out/${T}_sac_stamps_128.txt 2>&1
timeout -k 10 200 python scripts/rollout_lean_dev.py > gpurun_out/${T}_rollout_lean_stamps.txt 2>&1
timeout -k 10 200 python scripts/rollout_phase_stamps.py > gpurun_out/${T}_rollout_stamps.txt 2>&1
timeout -k 10 200 python scripts/bptt_op_stamps.py > gpurun_out/${T}_bptt_op_stamps.txt 2>&1
timeout -k 10 200 python scripts/layered_timing.py > gpurun_out/${T}_layered_timing.txt 2>&1
timeout -k 10 200 python scripts/layered_ppo_timing.py > gpurun_out/${T}_layered_ppo_timing.txt 2>&1
tail -3 gpurun_out/${T}_sac_lean_stamps.txt
