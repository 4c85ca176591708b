"""BASELINE configs[2] (PPO, N=16384, B=512, M=32, T from argv) issued eagerly for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ppo_prof -- python3 scripts/ppo_c3_profile.py 40"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))
import torch
import bench
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
orig_capturable = None
from mbpo.optimizers.policy_optimizers.ppo import ppo as ppo_mod
ppo_mod.PPO._capturable = lambda self: False          # eager: every launch is a dispatch the profiler sees
out = bench.ppo_c3_extra(dev, T, steps=3)
print(out)
