#!/usr/bin/env python
"""bench.py — MBPO inner loop on MI355X: model-rollout transitions/s (+ SAC updates/s).

Workload = BASELINE.json configs[1] ("Pendulum SAC-MBPO, 4096 parallel envs, 5-ens, horizon-5 rollouts, 1xMI355X") with
the north_star's obs/act shape: x=4, u=1, 5-member ensemble [5->64->64->64->8] swish, policy/twin-Q 64x3 swish,
N=4096 envs PER GPU, episode_length=5, num_env_steps_between_updates S=5, batch_size B=256 per GPU,
grad_updates_per_step G=64, normalize_observations=True, synthetic data, random-init weights.

One "step" = one SAC.training_step (sac/sac.py:306-327): fused rollout of S*N transitions + running statistics +
replay insert + sample of B*G rows + G sgd_steps (each: fwd/bwd of the three losses, [all-reduce,] clip+AdamW+Polyak).
value = transitions written per second by the whole job = n_gpus * N * S * steps / time  (weak scaling: per-GPU work fixed).

    python bench.py [--gpus N --steps K --warmup W]      # N > 1 without WORLD_SIZE in the environment: bench.py starts its N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

After the headline measurement rank 0 (single-GPU run only) also measures, and appends as extra keys of the same JSON line:
  steady_ms_per_step   the same hipGraph replayed >= 500 times (the timed region of the contract is K steps = tens of ms)
  ppo_c3               BASELINE configs[2]: PPO.training_step at N=16384, B=512, M=32, T=40 and T=5 (ppo/ppo.py:179-247)
  bptt_c5              BASELINE configs[4] per-GPU shape: BPTTOptimizer train step, E=10, H=32, x=17, u=6, n=4096 (bptt_optimizer.py:355-437)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))

import torch  # noqa: E402

X_DIM, U_DIM, N_MEMBERS = 4, 1, 5
HIDDEN = (64, 64, 64)
N_ENVS, EPISODE_LEN, S_STEPS, BATCH, GRAD_UPDATES = 4096, 5, 5, 256, 64
MAX_REPLAY = 2 ** 20
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak (spec)


def mlp_macs(dims):
    return sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))


def build_trainer(device, process_group, use_graph):
    from mbpo.optimizers.policy_optimizers.sac.sac import SAC
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import EnsembleDynamics, EnsembleSystem, QuadraticReward
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition

    dyn = EnsembleDynamics(X_DIM, U_DIM, n_members=N_MEMBERS, hidden_layer_sizes=HIDDEN, device=device)
    system = EnsembleSystem(dyn, QuadraticReward(X_DIM, U_DIM), mode="mean", predict_delta=True)
    sys_params = system.init_params(1)
    # damp the random ensemble so 5-step rollouts stay O(1) (random-init lecun nets are ~unit gain)
    sys_params.dynamics_params.params.mul_(0.5)
    # synthetic TRUE buffer: 2^16 rows of (obs, act, reward, discount, next_obs)  
    g = torch.Generator().manual_seed(0)
    n_true = 2 ** 16
    th = (torch.rand(n_true, generator=g) * 2 - 1) * 3.14159265
    obs = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(n_true, generator=g) * 2 - 1) * 8,
                       (torch.rand(n_true, generator=g) * 2 - 1) * 8], dim=1)
    act = torch.rand(n_true, U_DIM, generator=g) * 2 - 1
    dummy = Transition(observation=torch.zeros(X_DIM), action=torch.zeros(U_DIM), reward=torch.zeros(1),
                       discount=torch.zeros(1), next_observation=torch.zeros(X_DIM))
    true_buffer = UniformSamplingQueue(n_true, dummy, 1, device=device)
    tbs = true_buffer.init(0)
    rows = torch.cat([obs, act, torch.zeros(n_true, 1), torch.ones(n_true, 1), obs], dim=1).to(device)
    tbs = true_buffer.insert_rows(tbs, rows)
    env = BraxWrapper(system, sys_params, tbs, true_buffer)
    steps_per_train = N_ENVS * S_STEPS
    trainer = SAC(environment=env, num_timesteps=steps_per_train * 1000, episode_length=EPISODE_LEN,
                  num_env_steps_between_updates=S_STEPS, num_envs=N_ENVS, batch_size=BATCH,
                  grad_updates_per_step=GRAD_UPDATES, normalize_observations=True, discounting=0.99, lr_policy=3e-4,
                  lr_q=3e-4, lr_alpha=3e-4, min_replay_size=steps_per_train, max_replay_size=MAX_REPLAY,
                  policy_hidden_layer_sizes=HIDDEN, critic_hidden_layer_sizes=HIDDEN, use_graph=use_graph,
                  process_group=process_group)
    return trainer


def time_kernel_alone(trainer, reps=200):
    """Average duration of the dominant kernel (k_sac_lean<4>: the SAC forward/backward launch specialised for the benchmark
    networks, csrc/sac_lean.hip) from HIP events on the launch stream."""
    import ctypes as C
    from mbpo import _hip
    lib = _hip.load()
    up = trainer.updater
    d = up.desc
    d.batch = trainer._batch_rows.data_ptr()
    st = torch.cuda.current_stream()
    for _ in range(10):
        _hip.check(lib.mbpo_sac_grads_phase(C.byref(d), 1, st.cuda_stream), "fwd_bwd")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        _hip.check(lib.mbpo_sac_grads_phase(C.byref(d), 1, st.cuda_stream), "fwd_bwd")
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def time_rollout_alone(trainer, ts, env_state, buffer_state, reps=30):
    st = torch.cuda.current_stream()
    from mbpo import ops
    spec = trainer.env.system.rollout_spec(env_state.system_params, trainer.device)
    nm, ns = trainer._norm(ts.normalizer_params)

    def run():
        ops.model_rollout(policy_params=ts.policy_params, policy_spec=trainer.policy_spec, x_dim=X_DIM, u_dim=U_DIM,
                          obs=env_state.obs, first_obs=env_state.info['first_obs'], steps=env_state.info['steps'],
                          done=env_state.done, n_steps=S_STEPS, episode_length=EPISODE_LEN, norm_mean=nm, norm_std=ns,
                          seed=3, out=trainer._rollout_rows, **spec)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def cpu_baseline(budget_s=12.0):
    """The oracle's torch-CPU restatement of the same training_step, timed on this box's host cores."""
    from oracle import nets, sac as osac, systems as osys, trainer as otr
    g = torch.Generator().manual_seed(0)
    dd = [X_DIM + U_DIM, *HIDDEN, 2 * X_DIM]
    dpar = torch.cat([nets.init_mlp_flat(dd, g) * 0.5 for _ in range(N_MEMBERS)])
    rf = lambda x, u: osys.quadratic_reward(x, u, torch.zeros(X_DIM), torch.ones(X_DIM), torch.ones(U_DIM) * 0.1)
    system = osys.EnsembleSystem(dpar, dd, N_MEMBERS, X_DIM, U_DIM, reward_fn=rf)
    cfg = osac.SacConfig(X_DIM, U_DIM, [X_DIM, *HIDDEN, 2 * U_DIM], [X_DIM + U_DIM, *HIDDEN, 1], discounting=0.99,
                         lr_policy=3e-4, lr_q=3e-4, lr_alpha=3e-4)
    # the GPU box exposes the whole host in os.cpu_count() but gives this job a ~16-core share: more threads than that
    # only makes torch's CPU kernels spin against each other
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))
    log(f"cpu baseline on {torch.get_num_threads()} threads (affinity {avail}, cpu_count {os.cpu_count()})")
    loop = otr.CpuSacLoop(cfg, system, N_ENVS, S_STEPS, EPISODE_LEN, BATCH, GRAD_UPDATES, MAX_REPLAY, True)
    loop.training_step(n_sgd=2)   # warm-up
    t0 = time.time()
    n = 0
    while (time.time() - t0 < budget_s or n == 0) and n < 64:
        loop.training_step()
        n += 1
        log(f"cpu baseline: {n} training_steps in {time.time() - t0:.1f} s")
    dt = time.time() - t0
    return {"value": N_ENVS * S_STEPS * n / dt, "unit": "transitions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full SAC training_steps (N={N_ENVS}, S={S_STEPS}, G={GRAD_UPDATES}, B={BATCH}) of oracle/trainer.py "
                      f"(torch-CPU restatement, not XLA-CPU) in {dt:.1f} s",
            "sac_updates_per_s": GRAD_UPDATES * n / dt}


def _events_ms(fn, reps):
    """Device time of `reps` calls of fn() on the current stream, ms per call (HIP events on the launch stream)."""
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ppo_c3_extra(device, unroll_length: int, steps: int = 10):
    """BASELINE configs[2]: PPO.training_step (ppo/ppo.py:179-233) at N = 16384 envs, B = 512, M = 32 (B*M = N: one
    unroll per step), num_updates_per_batch = 8, gamma .99, lambda .95, eps .3, 64x3 nets, Pendulum model; one hipGraph replay per
    step (the reference compiles the epoch scan into one XLA computation, ppo.py:235-247).  GAE elements per step = E*M*B*T: every
    minibatch_step runs compute_gae (ppo/losses.py:128-184) over its [T, B] block."""
    from mbpo.optimizers.policy_optimizers.ppo.ppo import PPO
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition
    N, B, M, E, T = 16384, 512, 32, 8, unroll_length
    system = PendulumSystem()
    sp = system.init_params(1)
    g = torch.Generator().manual_seed(0)
    n_true = 2 ** 16
    th = (torch.rand(n_true, generator=g) * 2 - 1) * 3.14159265
    obs = torch.stack([torch.cos(th), torch.sin(th), (torch.rand(n_true, generator=g) * 2 - 1) * 8], dim=1)
    act = torch.rand(n_true, 1, generator=g) * 2 - 1
    dummy = Transition(observation=torch.zeros(3), action=torch.zeros(1), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(3))
    tb = UniformSamplingQueue(n_true, dummy, 1, device=device)
    tbs = tb.insert_rows(tb.init(0), torch.cat([obs, act, torch.zeros(n_true, 1), torch.ones(n_true, 1), obs], dim=1).to(device))
    env = BraxWrapper(system, sp, tbs, tb)
    tr = PPO(environment=env, num_timesteps=B * M * T * 1000, episode_length=200, num_envs=N, unroll_length=T, batch_size=B,
             num_minibatches=M, num_updates_per_batch=E, normalize_observations=True, discounting=0.99, gae_lambda=0.95,
             clipping_epsilon=0.3, entropy_cost=1e-2, lr=3e-4, reward_scaling=1.0, policy_hidden_layer_sizes=HIDDEN,
             critic_hidden_layer_sizes=HIDDEN)
    ts = tr.init_training_state(7)
    es = env.reset([1000 + i for i in range(N)])
    tr.rekey(23)
    ts, es, _ = tr.training_step(ts, es)            # eager: warms every kernel
    torch.cuda.synchronize()
    graph = None
    if tr._capturable():
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tr.training_step(ts, es)
        graph.replay()

    def step():
        if graph is not None:
            graph.replay()
        else:
            tr.training_step(ts, es)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    finite = bool(torch.isfinite(tr.updater.params).all())
    launches = 1 + 3 + E * (2 + 1 + M * 6) + 1
    tr.close()
    return {"N": N, "B": B, "M": M, "T": T, "num_updates_per_batch": E, "ms_per_step": ms, "graph": graph is not None,
            "gae_elements_per_s": E * M * B * T / (ms * 1e-3), "transitions_per_s": B * M * T / (ms * 1e-3),
            "minibatch_steps_per_s": E * M / (ms * 1e-3), "params_finite": finite, "steps_timed": steps,
            "approx_launches_per_step": launches}


def bptt_c5_extra(device, steps=(6, 26)):
    """BASELINE configs[4] at its per-GPU shape: BPTTOptimizer train steps (bptt_optimizer.py:355-437: actor gradient
    through a 10-member ensemble over H = 32 steps from n = 4096 initial states, AdamW, one critic regression update, normalisers,
    buffer insert), x = 17, u = 6, 64x3 nets.  Timed through the public train(): two calls with different train_steps, the
    difference divided by the difference in steps (set-up cancels).  (state x step)/s = n*H per train step."""
    from mbpo.optimizers import BPTTOptimizer
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import EnsembleDynamics, EnsembleSystem, QuadraticReward
    from mbpo.types import Transition
    X, U, E, H, n = 17, 6, 10, 32, 4096
    dyn = EnsembleDynamics(X, U, n_members=E, hidden_layer_sizes=HIDDEN, device=device)
    system = EnsembleSystem(dyn, QuadraticReward(X, U), mode="mean", predict_delta=True)
    g = torch.Generator().manual_seed(0)
    rows = 2 ** 14
    obs = torch.randn(rows, X, generator=g)
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    tb = UniformSamplingQueue(rows, dummy, 1, device=device)
    tbs = tb.insert_rows(tb.init(0), torch.cat([obs, torch.zeros(rows, U), torch.zeros(rows, 1), torch.ones(rows, 1), obs], dim=1).to(device))
    times = {k: float("inf") for k in steps}
    # Each length three times, the fastest run counts, and every run starts from the same allocator state: the actor kernel's workspace
    # is 1 GB at this shape (the members' pre-activations), and a run that has to hipMalloc it while the other length reuses a cached
    # block biased the difference (one bench run printed 2.96 ms per step against 4.0 ms for the actor gradient alone).
    import gc
    for k in (*steps, *steps, *steps):
        gc.collect()
        torch.cuda.empty_cache()
        opt = BPTTOptimizer(action_dim=U, obs_dim=X, horizon=H, num_samples_per_gradient_update=n, train_steps=k,
                            critic_updates_per_policy_update=1, sampling_buffer_size=rows + (max(steps) + 2) * n * H)
        opt.set_system(system)
        st = opt.init(key=5, true_buffer_state=tbs)
        st.system_params.dynamics_params.params.mul_(0.5)        # keep 32-step rollouts of a random ensemble O(1), as the headline does
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = opt.train(st)
        torch.cuda.synchronize()
        times[k] = min(times[k], time.perf_counter() - t0)
        finite = bool(torch.isfinite(out.optimizer_state.actor_params).all())
        del opt, out
    ms = (times[steps[1]] - times[steps[0]]) / (steps[1] - steps[0]) * 1e3
    return {"E": E, "H": H, "x": X, "u": U, "n": n, "ms_per_train_step": ms, "state_steps_per_s": n * H / (ms * 1e-3),
            "params_finite": finite, "graph": True, "timed": f"train(train_steps={steps[1]}) - train(train_steps={steps[0]})"}


def count_gpus_without_runtime():
    """Visible GPUs without any HIP call: the *_VISIBLE_DEVICES lists if set, else the KFD topology nodes that have SIMDs
    (/sys/class/kfd/kfd/topology/nodes/*/properties: CPUs report simd_count 0).  None when neither source is readable — the ranks
    then fail by themselves with a clear message if a device is missing."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    try:
        cnt = 0
        for node in sorted(Path("/sys/class/kfd/kfd/topology/nodes").iterdir()):
            for line in (node / "properties").read_text().splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    cnt += 1
        return cnt
    except (OSError, ValueError):
        return None


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` (N > 1) as ONE command: start N fresh rank processes of this same file — RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set — relay rank 0's single JSON line and
    return non-zero if any rank failed.  This parent never touches the GPU: devices are counted from the environment / sysfs (on a
    ROCm build without amdsmi bindings torch.cuda.device_count() falls back to hipGetDeviceCount, which initialises the runtime:
    ADVICE r3), the ranks are fresh child processes."""
    import socket
    import subprocess
    share = os.environ.get("MBPO_BENCH_SHARE_GPU") == "1"
    n_dev = count_gpus_without_runtime()
    if n_dev is not None and n_dev < n and not share:
        print(f"bench.py --gpus {n}: only {n_dev} GPU(s) visible (MBPO_BENCH_SHARE_GPU=1 rehearses all ranks on one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = b""
    import threading
    def drain():
        nonlocal out0
        out0 = procs[0].stdout.read()
    th = threading.Thread(target=drain, daemon=True)
    th.start()
    rc, deadline = 0, None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad and deadline is None:            # one rank died: its peers would wait in a collective for ever
            deadline = time.time() + 20.0
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                    # exact PIDs this parent started
        time.sleep(0.2)
    th.join(timeout=5.0)
    codes = [p.returncode for p in procs]
    if any(codes):
        print(f"bench.py --gpus {n}: rank exit codes {codes}", file=sys.stderr)
        rc = next(c for c in codes if c) or 1
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        print(f"bench.py --gpus {n}: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    return rc


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ppo_c3 / bptt_c5 extra measurements")
    args = ap.parse_args()

    # Libraries (RCCL's version banner) write to fd 1; the contract is ONE JSON line on stdout, so fd 1 points at stderr
    # until the result is printed.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(saved_stdout, 1)
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the number of ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal only (one-GPU box): MBPO_BENCH_SHARE_GPU=1 puts every rank on cuda:0, MBPO_BENCH_BACKEND=gloo replaces RCCL
    share_gpu = os.environ.get("MBPO_BENCH_SHARE_GPU") == "1"
    # (RCCL refuses two ranks on one device — "Duplicate GPU detected" — so the shared-GPU rehearsal defaults to gloo)
    backend = os.environ.get("MBPO_BENCH_BACKEND", "gloo" if share_gpu else "nccl")
    dev_index = 0 if share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        # (the launcher's sysfs count can see more devices than this container may open)
        raise SystemExit(f"bench.py rank {rank}: only {torch.cuda.device_count()} GPU(s) visible "
                         "(MBPO_BENCH_SHARE_GPU=1 rehearses all ranks on one GPU)")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    pg = None
    force_pg = os.environ.get("MBPO_BENCH_FORCE_PG") == "1"      # exercise the RCCL path on a single rank
    if world > 1 or force_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if force_pg and world == 1:
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group(backend=backend, **({"device_id": device} if backend == "nccl" else {}))
        pg = dist.group.WORLD

    def measure():
        log(f"building trainer on {device} (world={world})")
        trainer = build_trainer(device, pg, use_graph=not args.no_graph)
        from mbpo.utils import keys as K
        ts = trainer.init_training_state(7)
        rk = trainer.dp.rank_key       # per-rank data keys, as SAC.run_training derives them
        env_state = trainer.reset_envs(trainer.env, rk(11), N_ENVS)
        buffer_state = trainer.replay_buffer.init(rk(13))
        ts, env_state, buffer_state, _ = trainer.prefill_replay_buffer(ts, env_state, buffer_state, rk(17))
        torch.cuda.synchronize()
        log("prefill done")

        use_graph = trainer.use_graph
        trainer.rekey(rk(23))      # seed word of the device RNG control; the step index counts on the device from here on

        def one_step():
            nonlocal ts, env_state, buffer_state
            ts, env_state, buffer_state = trainer.training_step(ts, env_state, buffer_state)

        graph = None
        # warm-up: W eager steps (also compiles/loads every kernel), then capture
        for _ in range(max(args.warmup, 1)):
            one_step()
        torch.cuda.synchronize()
        log(f"{max(args.warmup, 1)} eager warm-up steps done")
        # Collectives inside the captured step: the peer-memory exchange (plain kernels) and RCCL collectives (stream-ordered
        # kernels) are captured; a host-side collective (gloo) is NEVER put inside a capture — it invalidates the capture and
        # the process cannot recover from that (round 1: hipErrorStreamCaptureInvalidated, then SIGSEGV in the same process;
        # DESIGN §6) — such a step is issued eagerly.  The decision is the trainer's (SAC._capturable).
        if use_graph and not trainer._capturable():
            log("gradient exchange is a host-side collective: issuing the step eagerly (no hipGraph)")
            use_graph = False
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                trainer.training_step(ts, env_state, buffer_state)
            log("graph captured")
            graph.replay()   # one untimed replay
            torch.cuda.synchronize()
            log("graph replayed once")

        def barrier():
            if pg is not None:
                import torch.distributed as dist
                dist.barrier()

        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                one_step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if pg is not None:
            import torch.distributed as dist
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        if os.environ.get("MBPO_BENCH_CHUNKS") and graph is not None:
            # diagnostic (stderr only, after the timed region): is the step time stable within one process?
            per = []
            for _ in range(int(os.environ["MBPO_BENCH_CHUNKS"])):
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for _ in range(20):
                    graph.replay()
                torch.cuda.synchronize()
                per.append((time.perf_counter() - c0) / 20 * 1e3)
            log("chunks of 20 steps, ms/step: " + " ".join(f"{v:.3f}" for v in per))

        return trainer, ts, env_state, buffer_state, graph, dt

    for _ in range(int(os.environ.get("MBPO_BENCH_REMEASURE", "0"))):
        # diagnostic: does the step time depend on where THIS trainer instance's buffers landed?  (stderr only)
        tr_, *_rest, g_, dt_ = measure()
        log(f"extra trainer instance: {dt_ / args.steps * 1e3:.3f} ms/step  params@{tr_.updater.params.data_ptr():#x} "
            f"workspace@{tr_.updater.workspace.data_ptr():#x}")
        del tr_, _rest, g_
    trainer, ts, env_state, buffer_state, graph, dt = measure()
    if pg is not None and getattr(trainer, "p2p", None) is not None:
        # a peer exchange that timed out poisons the gradients with NaN (csrc/p2p.hpp); such a run is not a measurement:
        # every rank then repeats the whole run over the RCCL all-reduce
        import torch.distributed as dist
        bad = torch.tensor([int(trainer.p2p.status() != 0 or not bool(torch.isfinite(trainer.updater.params).all()))],
                           device=device, dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad):
            log("peer-memory exchange failed during the run; measuring again with torch.distributed all_reduce")
            os.environ["MBPO_P2P_ALLREDUCE"] = "0"
            trainer, ts, env_state, buffer_state, graph, dt = measure()

    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    # the contract's timed region is K steps (tens of ms); the same graph over >= 500 replays, device time — EVERY rank replays
    # (the captured step holds the gradient exchange), the slowest rank's figure is reported
    steady_ms, n_steady = None, max(500, args.steps)
    if graph is not None:
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
        steady_ms = _events_ms(graph.replay, n_steady)
        if pg is not None:
            t = torch.tensor([steady_ms], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            steady_ms = float(t)
        log(f"steady state over {n_steady} replays: {steady_ms:.4f} ms/step")
    finite = bool(torch.isfinite(trainer.updater.params).all())
    # which device every rank sits on, and — if the peer exchange was declined — why (for whoever reads an 8-GPU run's line)
    from mbpo.parallel import P2PExchange
    dev_ids, p2p_reasons = [torch.cuda.current_device()], [P2PExchange.last_decline_reason]
    replicas_identical = None
    if pg is not None:
        import torch.distributed as dist
        gathered = [None] * dist.get_world_size(pg)
        dist.all_gather_object(gathered, (torch.cuda.current_device(), os.getpid(), P2PExchange.last_decline_reason), group=pg)
        dev_ids = [g_[0] for g_ in gathered]
        p2p_reasons = [g_[2] for g_ in gathered]
        # data parallelism keeps the replicas bit-identical (every rank adds the same gradients in the same order): a checksum of
        # the parameter words per rank says whether the exchange really delivered the same sums everywhere
        sums = [None] * dist.get_world_size(pg)
        dist.all_gather_object(sums, int(trainer.updater.params.view(torch.int32).to(torch.int64).sum()), group=pg)
        replicas_identical = len(set(sums)) == 1
        if not replicas_identical:
            log(f"WARNING: parameter checksums differ across ranks: {sums}")
    if rank == 0:
        P = mlp_macs([X_DIM, *HIDDEN, 2 * U_DIM])
        Q = mlp_macs([X_DIM + U_DIM, *HIDDEN, 1])
        M = mlp_macs([X_DIM + U_DIM, *HIDDEN, 2 * X_DIM])
        flop_per_sample = 2 * (5 * P + 12 * Q)                 # per sgd_step sample
        flop_per_transition = 2 * N_MEMBERS * M + 2 * P        # per model-rollout transition
        t_kernel = time_kernel_alone(trainer)
        t_roll = time_rollout_alone(trainer, ts, env_state, buffer_state)
        achieved = BATCH * flop_per_sample / t_kernel / 1e12
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be collected from inside this process (separate
        # rocprofv3 --pmc passes, MI355X_MICROARCH.md); the committed measurement of this same command is reported — but only
        # while the kernel sources still hash to what it was collected on (scripts/make_pmc_traffic.py): a stale file is refused
        traffic, traffic_note = None, "no PMC measurement committed"
        try:
            sys.path.insert(0, str(ROOT / "scripts"))
            from make_pmc_traffic import source_sha16
            pmc_path = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))[-1]        # the newest round's collection
            pmc = json.loads(pmc_path.read_text())
            cand = {n: v for n, v in pmc["kernels"].items() if n.startswith("k_sac_lean<") or n.startswith("k_sac_fwd_bwd<64")}
            k = max(cand.values(), key=lambda v: v["dispatches"])     # the variant this workload launches
            if pmc.get("source_sha16") != source_sha16():
                traffic_note = (f"profiles/{pmc_path.name} was collected on other kernel sources (hash mismatch): refused; "
                                "re-run scripts/collect_pmc.sh + scripts/make_pmc_traffic.py")
            else:
                traffic = int((k["fetch_kb"] + k["write_kb"]) * 1024)
                traffic_note = (f"FETCH_SIZE + WRITE_SIZE per launch from profiles/{pmc_path.name} (rocprofv3 --pmc, separate passes; "
                                "source hash checked); the writes are the 16 per-tile gradient slabs the fixed-order cross-tile reduction "
                                "reads back, the reads the weights once per workgroup (48 workgroups, 8 L2s) + the tile rows")
        except Exception as e:      # noqa: BLE001
            traffic_note = f"no usable PMC measurement ({type(e).__name__})"
        out = {
            "metric": "model-rollout transitions/sec + SAC updates/sec, Pendulum 5-ens h=5",
            "value": world * N_ENVS * S_STEPS * args.steps / dt,
            "unit": "transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SAC-MBPO training_step, BASELINE configs[1]: x=4 u=1, 5-member ensemble 64x3, "
                                   "N=4096 envs/GPU, episode_length=5, S=5, B=256/GPU, G=64, nets 64x3, normalize_observations",
                       "n_envs_per_gpu": N_ENVS, "horizon": EPISODE_LEN, "env_steps_between_updates": S_STEPS,
                       "batch_size_per_gpu": BATCH, "grad_updates_per_step": GRAD_UPDATES, "ensemble": N_MEMBERS,
                       "hipgraph": graph is not None, "parallelism": f"dp{world} (envs+minibatch sharded, flat grad all-reduce per sgd_step)",
                       "grad_exchange": ("none" if world == 1 and pg is None else
                                         "peer-memory one-shot (xGMI stores, csrc/p2p.hpp)" if getattr(trainer, "p2p", None) is not None
                                         else "torch.distributed all_reduce"),
                       # diagnostics for the multi-GPU run: what the start-up check measured and what the process group is
                       "p2p_timing_ms": (dict(zip(("p2p", "library"), trainer.p2p.timing_ms))
                                         if getattr(trainer, "p2p", None) is not None and hasattr(trainer.p2p, "timing_ms") else None),
                       "p2p_fused_exchange": bool(getattr(trainer.updater, "p2p_fused", False)) if getattr(trainer, "p2p", None) is not None else None,
                       "pg_backend": (None if pg is None else __import__("torch.distributed").distributed.get_backend(pg)),
                       "pg_world_size": (None if pg is None else __import__("torch.distributed").distributed.get_world_size(pg)),
                       "rank_device_ids": dev_ids,
                       "replicas_identical": replicas_identical,
                       "p2p_decline_reason": (None if getattr(trainer, "p2p", None) is not None or pg is None else p2p_reasons)},
            "sac_updates_per_s": GRAD_UPDATES * args.steps / dt,
            "sac_update_samples_per_s": world * BATCH * GRAD_UPDATES * args.steps / dt,
            "params_finite": finite,
            "roofline": {"bound": "mfma", "kernel": "k_sac_lean<4,false>", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "avg_launch_us": t_kernel * 1e6, "algorithmic_flop_per_launch": BATCH * flop_per_sample,
                         "launches_per_step": GRAD_UPDATES},
            "rollout_kernel": {"kernel": "k_rollout_lean<4,false>", "avg_launch_us": t_roll * 1e6,
                               "transitions_per_s_alone": N_ENVS * S_STEPS / t_roll,
                               "achieved_tflops": N_ENVS * S_STEPS * flop_per_transition / t_roll / 1e12,
                               "frac_of_fp32_mfma_peak": N_ENVS * S_STEPS * flop_per_transition / t_roll / 1e12 / FP32_MFMA_PEAK_TFLOPS},
        }
        log("kernel timings done")
        if steady_ms is not None:
            out["steady_ms_per_step"] = steady_ms
            out["steady_replays"] = n_steady
            out["steady_transitions_per_s"] = world * N_ENVS * S_STEPS / (steady_ms * 1e-3)
        if world == 1 and pg is None and not args.no_extras:
            del graph
            trainer.close()
            try:
                out["ppo_c3"] = {"T40": ppo_c3_extra(device, 40), "T5": ppo_c3_extra(device, 5, steps=20)}
                log(f"ppo_c3: T=40 {out['ppo_c3']['T40']['ms_per_step']:.2f} ms/step, T=5 {out['ppo_c3']['T5']['ms_per_step']:.2f} ms/step")
            except Exception as e:      # noqa: BLE001 — an extra must never cost the headline line
                out["ppo_c3"] = {"error": repr(e)}
            try:
                out["bptt_c5"] = bptt_c5_extra(device)
                log(f"bptt_c5: {out['bptt_c5']['ms_per_train_step']:.2f} ms per train step")
            except Exception as e:      # noqa: BLE001
                out["bptt_c5"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
            log("cpu baseline done")
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
