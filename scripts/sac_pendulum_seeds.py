"""Diagnostic: the reference's SAC acceptance configuration (tests/test_sac.py:30-57) over several keys, on the
HIP path or on the CPU oracle loop driven by the SAME key derivation and Philox streams (so a key means the same experiment on
both).  Prints the eval curve and whether the reference's thresholds (eval >= -400, |final reward| <= 0.1) are met.

    python scripts/sac_pendulum_seeds.py hip 10        # on the GPU box
    python scripts/sac_pendulum_seeds.py oracle 10     # CPU, ~1-2 min per key
"""
import math, sys, time
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo.utils import keys as K

backend = sys.argv[1] if len(sys.argv) > 1 else "hip"
n_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 6
first_key = int(sys.argv[3]) if len(sys.argv) > 3 else 0
CFG = dict(num_timesteps=20_000, num_evals=20, reward_scaling=1, episode_length=200, normalize_observations=True, action_repeat=1,
           discounting=0.99, lr_policy=3e-4, lr_alpha=3e-4, lr_q=3e-4, num_envs=32, batch_size=64, grad_updates_per_step=20 * 32,
           max_replay_size=2 ** 14, min_replay_size=2 ** 7, num_eval_envs=1, deterministic_eval=True, tau=0.005, wd_policy=0, wd_q=0,
           wd_alpha=0, num_env_steps_between_updates=20, policy_hidden_layer_sizes=(128, 128, 128),
           critic_hidden_layer_sizes=(128, 128, 128))


def run_hip(key):
    from mbpo.optimizers import SACOptimizer
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.types import Transition
    dev = torch.device('cuda:0')
    system = PendulumSystem()
    s0 = system.reset()
    dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward, discount=torch.tensor(0.99, device=dev),
                       next_observation=s0.x_next)
    buf = UniformSamplingQueue(10, dummy, 1, device=dev)
    sbs = buf.insert(buf.init(0), Transition(observation=s0.x_next[None], action=torch.zeros(1, 1, device=dev), reward=s0.reward[None],
                                             discount=torch.tensor([0.99], device=dev), next_observation=s0.x_next[None]))
    opt = SACOptimizer(system=system, true_buffer=buf, **CFG)
    out = opt.train(opt.init(key=key, true_buffer_state=sbs))
    x, st, r = s0.x_next, out.optimizer_state, 0.0
    for _ in range(200):
        u, st = opt.act(x, st, evaluate=True)
        nxt = system.step(x, u, st.system_params)
        x, r = nxt.x_next, float(nxt.reward)
    return [m["eval/episode_reward"] for m in out.summary], r


def run_oracle(key):
    """SACOptimizer.init/train + SAC.run_training's key structure (brax_optimizers.py:58-99, sac/sac.py:404-494) around CpuSacLoop."""
    from mbpo.systems.ensemble_system import lecun_uniform_flat
    from oracle import nets, rollout as oro, sac as osac, systems as osys, trainer as otr
    torch.set_num_threads(4)
    X, U = 3, 1
    pd, qd = [X, 128, 128, 128, 2 * U], [X + U, 128, 128, 128, 1]
    # BraxOptimizer.set_system consumed one split of the optimizer key; init(key): keys = split(key, 3) -> state key = keys[2]
    state_key = K.split(key, 3)[2]
    _, run_key = K.split(state_key)                        # train(): key, new_key = split(opt_state.key); run_training(key=new_key)
    key, subkey = K.split(run_key)
    kp, kq = K.split(subkey)
    pol = lecun_uniform_flat(pd, torch.Generator().manual_seed(kp % (2 ** 63)))
    gq = torch.Generator().manual_seed(kq % (2 ** 63))
    q = torch.cat([lecun_uniform_flat(qd, gq) for _ in range(2)])
    params = torch.cat([pol, q, torch.zeros(1)])
    key, rb_key, env_key, eval_key = K.split(key, 4)
    cfg = osac.SacConfig(X, U, pd, qd, discounting=0.99, lr_policy=3e-4, lr_q=3e-4, lr_alpha=3e-4)
    N = 32
    obs0 = torch.tensor([[-1.0, 0.0, 0.0]]).repeat(N, 1)    # every env draws the buffer's one row
    loop = otr.CpuSacLoop(cfg, osys.PendulumSystem(), N, 20, 200, 64, 640, 2 ** 14, True, init_params=params, init_obs=obs0)

    def evaluate():
        nm, ns = loop._norm()
        first = oro.EnvState(obs0[:1].clone(), obs0[:1].clone(), torch.zeros(1), torch.zeros(1))
        er, _ = oro.evaluate(loop.system, loop.state.params[:cfg.P], pd, first, 200, 1, "swish", nm, ns, deterministic=True)
        return float(er[0])

    evals = [evaluate()]
    key, prefill_key = K.split(key)
    loop.rekey(K.split(prefill_key)[0])
    for _ in range(4):
        loop.prefill_step()
    for _ in range(19):
        key, epoch_key = K.split(key)
        loop.rekey(epoch_key)
        for _ in range(2):
            loop.training_step()
        key, _ = K.split(key)
        evals.append(evaluate())
    nm, ns = loop._norm()
    x, r = obs0[:1].clone(), 0.0
    sysm = osys.PendulumSystem()
    for _ in range(200):
        a = torch.tanh(nets.mlp_forward(loop.state.params[:cfg.P], pd, nets.normalize(x, nm, ns))[:, :U])
        x, rr = sysm.step(x, a)
        r = float(rr[0])
    return evals, r


ok = 0
for key in range(first_key, first_key + n_keys):
    t = time.time()
    evals, r_final = (run_hip if backend == "hip" else run_oracle)(key)
    good = evals[-1] >= -400 and abs(r_final) <= 0.1
    ok += good
    print(f"{backend} key {key}: {time.time() - t:.1f}s  final eval {evals[-1]:.1f}  best {max(evals):.1f}  |r_200| {abs(r_final):.3f}  "
          f"{'PASS' if good else 'fail'}  curve {[round(e) for e in evals]}", flush=True)
print(f"{backend}: {ok}/{n_keys} keys meet the reference thresholds")
