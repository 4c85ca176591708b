"""Diagnostic: the reference's tests/test_sac.py configuration on the HIP path for several seeds."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo.optimizers import SACOptimizer
from mbpo.replay import UniformSamplingQueue
from mbpo.systems import PendulumSystem
from mbpo.types import Transition
dev = torch.device('cuda:0')
system = PendulumSystem()
s0 = system.reset()
dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward, discount=torch.tensor(0.99, device=dev), next_observation=s0.x_next)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    buf = UniformSamplingQueue(10, dummy, 1, device=dev)
    sbs = buf.insert(buf.init(0), Transition(observation=s0.x_next[None], action=torch.zeros(1, 1, device=dev), reward=s0.reward[None], discount=torch.tensor([0.99], device=dev), next_observation=s0.x_next[None]))
    opt = SACOptimizer(system=system, true_buffer=buf, num_timesteps=20_000, num_evals=20, reward_scaling=1, episode_length=200, normalize_observations=True, action_repeat=1, discounting=0.99, lr_policy=3e-4, lr_alpha=3e-4, lr_q=3e-4, num_envs=32, batch_size=64, grad_updates_per_step=20 * 32, max_replay_size=2 ** 14, min_replay_size=2 ** 7, num_eval_envs=1, deterministic_eval=True, tau=0.005, num_env_steps_between_updates=20, policy_hidden_layer_sizes=(128, 128, 128), critic_hidden_layer_sizes=(128, 128, 128))
    t = time.time()
    out = opt.train(opt.init(key=seed, true_buffer_state=sbs))
    print(seed, round(time.time() - t, 1), [round(m['eval/episode_reward']) for m in out.summary], flush=True)
