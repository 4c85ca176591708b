"""Per-kernel register / spill table from the -Rpass-analysis=kernel-resource-usage dump of `build.py --save-temps`
(/tmp/mbpo_hip_build/resource_usage.txt): python scripts/resource_table.py [substring ...]"""
import re
import subprocess
import sys
from pathlib import Path

txt = Path("/tmp/mbpo_hip_build/resource_usage.txt").read_text()
blocks = re.findall(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)"
                    r".*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", txt, re.S)
names = subprocess.run(["c++filt"], input="\n".join(b[0] for b in blocks), capture_output=True, text=True).stdout.splitlines()
keys = sys.argv[1:] or ["k_sac_fwd_bwd", "k_sac_lean", "k_ppo_fwd_bwd", "k_ppo_values_gae", "k_bptt_actor", "k_model_rollout64", "k_rollout_lean", "k_critic_fwd",
                        "k_ensemble_forward", "k_mlp_vjp", "k_ens_nll"]
print(f"{'kernel':72s} {'VGPR':>4} {'spill':>5} {'scratch':>7} {'SGPR':>4} {'spill':>5} {'waves/SIMD':>10}")
for dem, (_, sg, vg, scr, occ, ssp, vsp) in zip(names, blocks):
    if any(k in dem for k in keys):
        print(f"{dem[:72]:72s} {vg:>4} {vsp:>5} {scr:>7} {sg:>4} {ssp:>5} {occ:>10}")
