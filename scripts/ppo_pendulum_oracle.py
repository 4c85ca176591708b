"""Diagnostic: the reference's PPO acceptance configuration (tests/test_ppo.py:30-56) on the CPU oracle loop
(oracle/trainer.py:CpuPpoLoop), with the key derivation and Philox streams of the HIP trainer: key k here is the same
experiment as key k of scripts/ppo_pendulum_seeds.py.   python scripts/ppo_pendulum_oracle.py <n_keys> <num_timesteps> [first_key]"""
import math, sys, time
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo.systems.ensemble_system import lecun_uniform_flat
from mbpo.utils import keys as K
from oracle import nets, ppo as oppo, rollout as oro, systems as osys, trainer as otr

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
first_key = int(sys.argv[3]) if len(sys.argv) > 3 else 0
torch.set_num_threads(4)
X, U, N, T, B, M, E = 3, 1, 256, 40, 128, 32, 8
pd, vd = [X, 64, 64, 2 * U], [X, 64, 64, 1]
per_epoch = math.ceil(steps / (19 * B * T * M))
for key in range(first_key, first_key + n_keys):
    t0 = time.time()
    state_key = K.split(key, 3)[2]                         # BraxOptimizer.init: keys = split(key, 3); state key = keys[2]
    _, run_key = K.split(state_key)                        # train(): run_training(key=new_key)
    k, subkey = K.split(run_key)
    k0, k1 = K.split(subkey)
    params = torch.cat([lecun_uniform_flat(pd, torch.Generator().manual_seed(k0 % (2 ** 63))),
                        lecun_uniform_flat(vd, torch.Generator().manual_seed(k1 % (2 ** 63)))])
    k, rb_key, env_key, eval_key = K.split(k, 4)
    cfg = oppo.PpoConfig(X, U, pd, vd, entropy_cost=1e-1, discounting=0.99, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.3,
                         normalize_advantage=True, lr=3e-3, wd=0.0)
    obs0 = torch.tensor([[-1.0, 0.0, 0.0]]).repeat(N, 1)
    loop = otr.CpuPpoLoop(cfg, osys.PendulumSystem(), N, T, 200, B, M, E, True, init_params=params, init_obs=obs0)

    def evaluate():
        nm, ns = loop._norm()
        first = oro.EnvState(obs0[:1].clone(), obs0[:1].clone(), torch.zeros(1), torch.zeros(1))
        return float(oro.evaluate(loop.system, loop.state.params[:cfg.P], pd, first, 200, 1, "swish", nm, ns, deterministic=True)[0][0])

    evals = [evaluate()]
    k, prefill_key = K.split(k)
    for _ in range(19):
        k, epoch_key = K.split(k)
        loop.rekey(epoch_key)
        for _ in range(per_epoch):
            loop.training_step()
        evals.append(evaluate())
    print(f"oracle ppo key {key}: {time.time() - t0:.0f}s  final eval {evals[-1]:.1f}  best {max(evals):.1f}  curve {[round(e) for e in evals]}", flush=True)
