"""Per-update time of the SAC step on the layered path (csrc/sac_layered.hip) next to the fused kernels, B = 256: a hipGraph of 32
chained mbpo_sac_step calls, replayed 20 times.  python scripts/layered_timing.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))
from mbpo import ops  # noqa: E402


def run(hidden_p, hidden_q, B=256, X=4, U=1, n=32):
    dev = torch.device("cuda", 0)
    up = ops.SacUpdater(x_dim=X, u_dim=U, policy_dims=[X, *hidden_p, 2 * U], q_dims=[X + U, *hidden_q, 1], batch_size=B, device=dev)
    g = torch.Generator().manual_seed(0)
    params = 0.05 * torch.randn(up.NP, generator=g)
    up.load_state(params.to(dev))
    batch = torch.randn(B, 2 * X + U + 3, generator=g).to(dev)
    rng = ops.make_rng(dev, 1, 0)
    for _ in range(3):
        up.sgd_step(batch, offset=0, rng_dev=rng, defer_clip_check=True)
    up.finalize()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(n):
            up.sgd_step(batch, offset=i, rng_dev=rng, defer_clip_check=True)
        up.finalize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * n)
    flop = 2 * B * (5 * ops.MlpSpec([X, *hidden_p, 2 * U]).n_params + 12 * ops.MlpSpec([X + U, *hidden_q, 1]).n_params)
    print(f"policy {hidden_p} critic {hidden_q} B={B}: {us:8.1f} us per update, {flop / us / 1e6:6.2f} TFLOP/s algorithmic, finite={bool(torch.isfinite(up.params).all())}")


if __name__ == "__main__":
    run((64, 64, 64), (64, 64, 64))
    run((128, 128, 128), (128, 128, 128))
    run((60, 60, 60), (60, 60, 60))            # layered at the fused kernel's size
    run((256, 256, 256), (256, 256, 256))
    run((256,) * 5, (256,) * 5)
    run((512, 512), (512, 512))
    run((256, 256, 256), (256, 256, 256), B=4096)
