"""Diagnostic: the reference's tests/test_ppo.py configuration (4M steps) on the HIP path for several keys:
last eval reward and the reward of the last step of a 200-step closed loop."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo.optimizers import PPOOptimizer
from mbpo.replay import UniformSamplingQueue
from mbpo.systems import PendulumSystem
from mbpo.types import Transition
dev = torch.device('cuda:0')
system = PendulumSystem()
s0 = system.reset()
dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward, discount=torch.tensor(0.99, device=dev), next_observation=s0.x_next)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
for key in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    buf = UniformSamplingQueue(10, dummy, 1, device=dev)
    sbs = buf.insert(buf.init(0), Transition(observation=s0.x_next[None], action=torch.zeros(1, 1, device=dev), reward=s0.reward[None], discount=torch.tensor([0.99], device=dev), next_observation=s0.x_next[None]))
    opt = PPOOptimizer(system=system, true_buffer=buf, num_timesteps=steps, episode_length=200, action_repeat=1, num_envs=256, num_eval_envs=1, lr=3e-3, wd=0, entropy_cost=1e-1, discounting=0.99, seed=0, unroll_length=40, batch_size=128, num_minibatches=32, num_updates_per_batch=8, num_evals=20, normalize_observations=True, reward_scaling=1, clipping_epsilon=0.3, gae_lambda=0.95, deterministic_eval=True, normalize_advantage=True, policy_hidden_layer_sizes=(64, 64), critic_hidden_layer_sizes=(64, 64))
    t = time.time()
    out = opt.train(opt.init(key=key, true_buffer_state=sbs))
    dt = time.time() - t
    x, st, r = s0.x_next, out.optimizer_state, 0.0
    for _ in range(200):
        u, st = opt.act(x, st, evaluate=True)
        nxt = system.step(x, u, st.system_params)
        x, r = nxt.x_next, float(nxt.reward)
    print(f"hip ppo key {key}: {dt:.1f}s  final eval {out.summary[-1]['eval/episode_reward']:.1f}  |r_200| {abs(r):.3f}  curve "
          f"{[round(m['eval/episode_reward']) for m in out.summary]}", flush=True)
