"""Random shapes through the layered SAC path against the fp64 oracle (hidden sizes 1..513, depths 1..4, B 1..2049, x 1..33, u 1..7):
    python scripts/fuzz_layered.py        (24 configurations; every one within 3e-6 + 3e-4 * max|g| on the round-3 library)"""
import sys, random
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch
import test_gpu_sac as T
from oracle import sac as osac
dev = torch.device("cuda", 0)
random.seed(1)
bad = 0
for it in range(24):
    X = random.choice([1, 2, 3, 5, 9, 17, 33]); U = random.choice([1, 2, 3, 7])
    hp = tuple(random.choice([1, 7, 31, 65, 130, 257, 300]) for _ in range(random.choice([1, 2, 3])))
    hq = tuple(random.choice([3, 16, 63, 129, 200, 513]) for _ in range(random.choice([1, 2, 4])))
    B = random.choice([1, 2, 15, 33, 64, 257, 1100, 2049])
    cfg, st, batch, noise, nm, ns = T._make(X, U, hp, B, 100 + it, True, q_hidden=hq)
    g_ref, (cl, ac, al) = osac.grads(cfg, st.params.double(), st.target_q.double(), batch.double(), *[n.double() for n in noise], nm.double(), ns.double())
    up = T._updater(dev, cfg, B, two_launch=False)
    up.load_state(st.params.to(dev), st.target_q.to(dev))
    up.sgd_step(batch.to(dev), nm.to(dev), ns.to(dev), *[n.to(dev) for n in noise])
    torch.cuda.synchronize()
    g = up.grads.cpu().double()
    err = float((g - g_ref).abs().max()); scale = float(g_ref.abs().max())
    ok = err <= 3e-6 + 3e-4 * scale
    bad += (not ok)
    print(f"{it:2d} X={X} U={U} hp={hp} hq={hq} B={B}: max err {err:.2e} (scale {scale:.2e}) {'ok' if ok else 'BAD'}", flush=True)
print("bad:", bad)
