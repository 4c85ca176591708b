"""BASELINE configs[4] at its per-GPU shape (BPTT, E = 10, H = 32, x = 17, u = 6, n = 4096) issued eagerly for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bptt_prof -- python3 scripts/bptt_c5_profile.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))
import torch
import bench
torch.cuda.set_device(0)
from mbpo.optimizers.policy_optimizers import bptt_optimizer as bo
orig = bo.BPTTOptimizer.__init__
def init(self, *a, **k):
    k["use_graph"] = False          # eager: every launch is a dispatch the profiler sees
    orig(self, *a, **k)
try:
    bo.BPTTOptimizer.__init__ = init
    print(bench.bptt_c5_extra(torch.device("cuda:0"), steps=(4, 12)))
finally:
    bo.BPTTOptimizer.__init__ = orig
