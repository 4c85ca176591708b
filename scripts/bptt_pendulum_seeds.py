"""Diagnostic: the reference's tests/test_bptt.py configuration on the HIP path for several keys."""
import math, sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'model-based-policy-optimizers_amd')
from mbpo.optimizers import BPTTOptimizer
from mbpo.replay import UniformSamplingQueue
from mbpo.systems import PendulumSystem
from mbpo.types import Transition
dev = torch.device('cuda:0')
system = PendulumSystem()
s0 = system.reset()
dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward, discount=torch.tensor(0.99, device=dev), next_observation=s0.x_next)
obs = torch.tensor([[math.cos(math.pi), math.sin(math.pi), 0.0]], device=dev)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    buf = UniformSamplingQueue(10000, dummy, 1, device=dev)
    sbs = buf.insert(buf.init(0), Transition(observation=obs, action=torch.zeros(1, 1, device=dev), reward=torch.zeros(1, device=dev), discount=torch.ones(1, device=dev), next_observation=obs))
    opt = BPTTOptimizer(action_dim=1, obs_dim=3, horizon=20, num_samples_per_gradient_update=50, train_steps=steps, init_stddev=2.0, lambda_=0.97, critic_updates_per_policy_update=1, use_best_trained_policy=True, sampling_buffer_size=2_000_000)
    opt.set_system(system)
    t = time.time()
    out = opt.train(bptt_state=opt.init(key=seed, true_buffer_state=sbs))
    torch.cuda.synchronize(); dt = time.time() - t
    st, x, tot = out.optimizer_state, s0.x_next, 0.0
    for _ in range(200):
        u, st = opt.act(obs=x, opt_state=st)
        nxt = system.step(x=x, u=u, system_params=st.system_params)
        x = nxt.x_next; tot += float(nxt.reward)
    al = out.bptt_summary.actor_loss
    print(seed, round(dt, 2), 's  closed-loop reward', round(tot, 1), ' actor_loss first/last', round(float(al[0]), 3), round(float(al[-1]), 3), flush=True)
