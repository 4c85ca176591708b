"""Diagnostic: per-layer cost of a wave chain = (t(deep) - t(shallow)) / extra layers, one 16-row tile per CU."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo import ops
dev = torch.device('cuda:0')
def t_fwd(dims, E, N, reps=200):
    spec = ops.MlpSpec(dims, 'swish', E)
    p = (torch.randn(spec.total_params) * 0.1).to(dev)
    x = torch.randn(N, dims[0]).to(dev)
    for _ in range(5): ops.ensemble_mlp_forward(p, spec, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.ensemble_mlp_forward(p, spec, x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for H in (64, 128):
    for E in (1, 4):
        for N in (16, 4096):
            a = t_fwd([H] + [H] * 2 + [H], E, N)
            b = t_fwd([H] + [H] * 6 + [H], E, N)
            print(f"H={H} E={E} N={N}: shallow(3 layers) {a:.1f} us, deep(7 layers) {b:.1f} us -> {(b - a) / 4:.2f} us/layer "
                  f"(MFMA floor {H*H*16*2/1024/ (64*2.4e3/32) * 1:.2f} us)", flush=True)
