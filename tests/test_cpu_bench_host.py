"""CPU: host-side logic of bench.py that runs before anything touches a GPU."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.mark.parametrize("env,want", [
    ({"HIP_VISIBLE_DEVICES": "0,1,2"}, 3),
    ({"HIP_VISIBLE_DEVICES": ""}, 0),
    ({"CUDA_VISIBLE_DEVICES": "3"}, 1),
    ({"ROCR_VISIBLE_DEVICES": "0, 1 ,2,3,4,5,6,7"}, 8),
    ({"HIP_VISIBLE_DEVICES": "0,1", "ROCR_VISIBLE_DEVICES": "0,1,2,3"}, 2),      # the HIP list narrows the ROCr one
])
def test_gpu_count_from_the_environment(monkeypatch, env, want):
    import bench
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert bench.count_gpus_without_runtime() == want


def test_bench_refuses_more_ranks_than_gpus_before_starting_any():
    """`python bench.py --gpus 4` with one visible GPU: exit code 2 and a message, no rank process started (nothing below the check
    can run here: there is no GPU in this container)."""
    import os
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    env.pop("WORLD_SIZE", None)
    env.pop("MBPO_BENCH_SHARE_GPU", None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    assert "only 1 GPU(s) visible" in p.stderr
