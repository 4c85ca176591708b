"""PPO minibatch_step on the layered path at BASELINE configs[2]'s minibatch (B = 512, T = 40 / 5) with the reference's
experiments/train_inverted_pendulum/exp_ppo.py networks (policy (32,)*4 padded to 64, critic (256,)*5), next to the fused 64x3 shape.
    python scripts/layered_ppo_timing.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))
from mbpo import ops  # noqa: E402


def run(hp, hv, B=512, T=40, X=3, U=1, n=16):
    dev = torch.device("cuda", 0)
    pd, vd = [X, *hp, 2 * U], [X, *hv, 1]
    up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=pd, value_dims=vd, batch_size=B, unroll_length=T, device=dev)
    g = torch.Generator().manual_seed(0)
    up.load_state((0.05 * torch.randn(up.params.numel(), generator=g)).to(dev))
    data = torch.randn(B, T, 2 * X + 2 * U + 4, generator=g)
    data[..., X + U + 1] = 1.0
    data[..., -1] = 0.0
    data = data.to(dev)
    rng = ops.make_rng(dev, 1, 0)
    for _ in range(2):
        up.minibatch_step(data, offset=0, rng_dev=rng)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(n):
            up.minibatch_step(data, offset=i, rng_dev=rng)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * n)
    M = B * T
    flop = 2 * M * (3 * ops.MlpSpec(pd).n_params + 4 * ops.MlpSpec(vd).n_params)
    print(f"policy {hp} value {hv} B={B} T={T}: {us:9.1f} us per minibatch step, {flop / us / 1e6:6.2f} TFLOP/s algorithmic, finite={bool(torch.isfinite(up.params).all())}")


if __name__ == "__main__":
    run((64, 64, 64), (64, 64, 64))
    run((64, 64, 64, 64), (256,) * 5)
    run((64, 64, 64, 64), (256,) * 5, T=5)
    run((256, 256), (256, 256))
