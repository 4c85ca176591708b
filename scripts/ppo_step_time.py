"""PPO minibatch_step launched repeatedly (for rocprofv3 --kernel-trace --stats): BASELINE config 3 shape B=512, T=5 / 40."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'model-based-policy-optimizers_amd'))
from mbpo import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
X, U, hid = 4, 1, (64, 64, 64)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def lecun(dims):
    parts = []
    for i, o in zip(dims[:-1], dims[1:]):
        parts += [((torch.rand(i, o, generator=g) * 2 - 1) * math.sqrt(3.0 / i)).reshape(-1), torch.zeros(o)]
    return torch.cat(parts)


pd, vd = [X, *hid, 2 * U], [X, *hid, 1]
up = ops.PpoUpdater(x_dim=X, u_dim=U, policy_dims=pd, value_dims=vd, batch_size=512, unroll_length=T, device=dev)
up.load_state(torch.cat([lecun(pd), lecun(vd)]).to(dev))
D = ops.transition_row_len(X, U, True)
data = torch.randn(512, T, D, generator=g) * 0.5
data[..., X + U + 1] = 1.0
data[..., -1] = 0.0
data = data.to(dev)
for _ in range(100):
    up.minibatch_step(data)
torch.cuda.synchronize()
