#!/bin/bash
# Copy what scripts/collect_profiles.sh + scripts/collect_round_extras.sh left under gpurun_out/ into profiles/ under a round tag
# (default r04).  Run in the dev container after the gpurun calls have merged gpurun_out/.
set -e
T=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
python scripts/make_pmc_traffic.py gpurun_out/pmc_fetch_size gpurun_out/pmc_write_size profiles/${T}_pmc_traffic.json > /dev/null
python scripts/make_pmc_sq.py gpurun_out/pmc_sq profiles/${T}_pmc_sq.json > /dev/null
python scripts/make_pmc_sq.py gpurun_out/ppo_c3_sq profiles/${T}_ppo_c3_pmc_sq.json "python3 scripts/ppo_c3_profile.py 40" > /dev/null
python scripts/make_pmc_sq.py gpurun_out/bptt_c5_sq profiles/${T}_bptt_c5_pmc_sq.json "python3 scripts/bptt_c5_profile.py" > /dev/null
cp gpurun_out/kernel_rooflines.json profiles/${T}_kernel_rooflines.json
cp $(ls -t gpurun_out/prof/runc/*_kernel_stats.csv | head -1) profiles/${T}_kernel_stats.csv
cp gpurun_out/prof_bench.json profiles/${T}_prof_bench.json
cp gpurun_out/bench.json profiles/${T}_bench.json
cp gpurun_out/${T}_ppo_c3_kernel_stats.csv gpurun_out/${T}_bptt_c5_kernel_stats.csv profiles/
mkdir -p profiles/pmc
cp $(ls -t gpurun_out/pmc_fetch_size/runc/*counter_collection.csv | head -1) profiles/pmc/${T}_fetch_size_counter_collection.csv
cp $(ls -t gpurun_out/pmc_write_size/runc/*counter_collection.csv | head -1) profiles/pmc/${T}_write_size_counter_collection.csv
cp $(ls -t gpurun_out/pmc_sq/runc/*counter_collection.csv | head -1) profiles/pmc/${T}_sq_counter_collection.csv
(echo "== scripts/lean_dev.py: k_sac_lean<4> against the generic k_sac_fwd_bwd<64,4,false,2,true> (bit identity, device time per launch / per two-launch update, s_memtime timeline of tile 0)"; grep -v amdgpu.ids gpurun_out/${T}_sac_lean_stamps.txt
 echo; echo "== scripts/ppo_lean_dev.py: k_ppo_lean<4> against the generic k_ppo_fwd_bwd<64,2> (BASELINE config 3 minibatch; per-tile timeline of workgroup 0)"; grep -v amdgpu.ids gpurun_out/${T}_ppo_lean_stamps.txt
 echo; echo "== scripts/sac_phase_stamps.py 128,128,128 (k_sac_fwd_bwd<128,4,false,2>)"; tail -22 gpurun_out/${T}_sac_stamps_128.txt
 echo; echo "== scripts/rollout_lean_dev.py: k_rollout_lean<4,false> against the generic k_model_rollout64 (device time per launch, bit identity, timeline of env step 1)"; grep -v amdgpu.ids gpurun_out/${T}_rollout_lean_stamps.txt
 echo; echo "== scripts/rollout_phase_stamps.py (the generic k_model_rollout64)"; grep -v amdgpu.ids gpurun_out/${T}_rollout_stamps.txt
 echo; echo "== scripts/bptt_op_stamps.py (k_bptt_actor, cycles per op kind)"; grep -v amdgpu.ids gpurun_out/${T}_bptt_op_stamps.txt
 echo; echo "== scripts/step_flavours.py (SAC update, one rank, per step flavour)"; grep -v amdgpu.ids gpurun_out/${T}_step_flavours.txt
 echo; echo "== scripts/layered_timing.py (SAC sgd_step per network shape; fused kernels and the layered path)"; grep -v amdgpu.ids gpurun_out/${T}_layered_timing.txt
 echo; echo "== scripts/layered_ppo_timing.py"; grep -v amdgpu.ids gpurun_out/${T}_layered_ppo_timing.txt) > profiles/${T}_phase_stamps.txt
tail -2 gpurun_out/tests_gpu.log
python - <<PY
import json
b = json.load(open("profiles/${T}_bench.json"))
print("bench:", round(b["value"] / 1e6, 3), "M transitions/s", round(b["ms_per_step"], 4), "ms/step; roofline", b["roofline"]["avg_launch_us"], "us frac", round(b["roofline"]["frac"], 4), "traffic", b["roofline"]["traffic"])
PY
