"""GPU end-to-end: the whole MBPO loop of examples/mbpo_pendulum.py on the HIP path — true Pendulum transitions, ensemble
fitting (N3), SAC on the learned EnsembleSystem (short model rollouts branched from true states), the resulting policy acting
on the TRUE system.  Thresholds are loose on purpose (one seed solves it to about -370..-390; an untrained policy sits near
-1500): the test guards the wiring between the stages, the arithmetic of each stage has its own parity test."""
import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.timeout(300)
def test_mbpo_pendulum_loop_improves_true_return(dev):
    spec = importlib.util.spec_from_file_location("mbpo_pendulum_example", ROOT / "examples" / "mbpo_pendulum.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.run(iters=2, verbose=False)
    assert len(hist) == 2 and hist[1]["true_transitions"] == 8000
    assert all(h["model_nll"] < -8.0 for h in hist), hist            # the ensemble fits the true dynamics tightly
    assert max(h["true_return"] for h in hist) >= -600.0, hist       # swing-up on the TRUE system from model-generated data only
