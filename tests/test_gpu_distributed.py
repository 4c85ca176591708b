"""GPU, world_size 2: two ranks share the one MI355X of the box (gloo moves the CUDA tensors), each runs the HIP SAC
sgd_step on HALF of a global minibatch with the flat-gradient all-reduce in between; the result must equal the
single-process HIP step on the whole minibatch.  Also a world_size-1 "nccl" (RCCL) group through the same code path."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _setup_paths():
    for p in (str(ROOT), str(ROOT / "model-based-policy-optimizers_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _make(world, B, hp=(64, 64, 64), hq=(64, 64, 64)):
    from oracle import sac as osac
    X, U = 4, 1
    g = torch.Generator().manual_seed(0)
    cfg = osac.SacConfig(X, U, [X, *hp, 2 * U], [X + U, *hq, 1], lr_policy=1e-3, lr_q=1e-3, lr_alpha=1e-3,
                         max_grad_norm=0.05)
    st = osac.init_state(cfg, g)
    D = 2 * X + U + 3
    batch = torch.randn(world * B, D, generator=g)
    batch[:, X + U + 1] = 1.0
    batch[:, D - 1] = (torch.rand(world * B, generator=g) < 0.2).float()
    noise = [torch.randn(world * B, U, generator=g) for _ in range(3)]
    return cfg, st, batch, noise


def _updater(cfg, B, dev, **kw):
    from mbpo import ops
    return ops.SacUpdater(x_dim=cfg.x_dim, u_dim=cfg.u_dim, policy_dims=cfg.policy_dims, q_dims=cfg.q_dims, batch_size=B,
                          device=dev, lr_policy=cfg.lr_policy, lr_q=cfg.lr_q, lr_alpha=cfg.lr_alpha,
                          max_grad_norm=cfg.max_grad_norm, **kw)


def _worker(rank, world, port, backend, tmpdir):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from mbpo.parallel import DataParallel
        dp = DataParallel(dist.group.WORLD)
        B = 32
        cfg, st, batch, noise = _make(world, B)
        sl = slice(rank * B, (rank + 1) * B)
        up = _updater(cfg, B, dev, all_reduce=dp.all_reduce_fn(), world_size=world)
        params = st.params.to(dev)
        dp.broadcast(params)
        up.load_state(params)
        up.sgd_step(batch[sl].to(dev), None, None, *[n[sl].to(dev) for n in noise])
        torch.cuda.synchronize()
        # single-process reference on the GLOBAL minibatch, same HIP path
        ref = _updater(cfg, world * B, dev)
        ref.load_state(st.params.to(dev))
        ref.sgd_step(batch.to(dev), None, None, *[n.to(dev) for n in noise])
        torch.cuda.synchronize()
        torch.testing.assert_close(up.grads * (1.0 / world), ref.grads, atol=1e-6, rtol=1e-4)
        torch.testing.assert_close(up.params, ref.params, atol=2e-6, rtol=0)
        torch.testing.assert_close(up.target_q, ref.target_q, atol=2e-6, rtol=0)
        (Path(tmpdir) / f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, "gloo", str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_single_rank_rccl_group(tmp_path):
    """world_size 1 over backend 'nccl' (= RCCL): the same all-reduce + grad-norm + apply sequence the 8-GPU run uses."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(1, port, "nccl", str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


def _p2p_worker(rank, world, port, tmpdir):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mbpo.parallel import DataParallel, P2PExchange
        dp = DataParallel(dist.group.WORLD)
        n = 26309
        ex = P2PExchange.create(dp, n, dev)
        assert ex is not None, "exchange regions / self-check failed"
        for it in range(5):
            g = torch.Generator().manual_seed(7 * it + rank)
            x = torch.randn(n, generator=g).to(dev)
            # expected: slots added in rank order, fp32
            exp = torch.zeros(n)
            for r in range(world):
                exp = exp + torch.randn(n, generator=torch.Generator().manual_seed(7 * it + r))
            ex.all_reduce_sum(x)
            torch.cuda.synchronize()
            assert ex.status() == 0
            assert torch.equal(x.cpu(), exp), f"iteration {it}"
        ex.close()
        (Path(tmpdir) / f"p2p_ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_p2p_all_reduce_two_ranks_one_gpu(tmp_path):
    """The one-shot peer-memory all-reduce (csrc/p2p.hpp) between two processes that share the box's GPU: regions exchanged by
    IPC handle, five exchanges, bit-exact rank-ordered fp32 sums on both ranks.  (Across GPUs the same stores travel over xGMI;
    P2PExchange.create re-validates against the library all-reduce at start-up and falls back to it otherwise.)"""
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_p2p_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"p2p_ok{r}").exists() for r in range(world))


def _p2p_sac_worker(rank, world, port, tmpdir, fused, layered=False, wide=False):
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mbpo.parallel import DataParallel, P2PExchange
        dp = DataParallel(dist.group.WORLD)
        B = 32
        cfg, st, batch, noise = _make(world, B, *(((64, 64), (256,) * 3) if wide else ((96, 40), (200,)) if layered else ()))
        sl = slice(rank * B, (rank + 1) * B)
        up = _updater(cfg, B, dev, world_size=world)
        ex = P2PExchange.create(dp, up.NP, dev)
        assert ex is not None
        up.p2p = ex
        if fused is not None:      # None: the updater's own default (fused where the reduction's workgroups can be co-resident)
            up.p2p_fused = fused
        if wide:
            assert up.p2p_fused and not up._p2p_fused_fits and (up.NP + 255) // 256 > 1024
        ref_rccl = _updater(cfg, B, dev, all_reduce=dp.all_reduce_fn(), world_size=world)     # library-collective path (gloo here)
        ref = _updater(cfg, world * B, dev)                                                   # single process, global minibatch
        for u in (up, ref_rccl, ref):
            u.load_state(st.params.to(dev))
        for it in range(4):       # chained steps: both slot parities, re-armed flags
            g = torch.Generator().manual_seed(100 + it)
            D = batch.shape[1]
            b = torch.randn(world * B, D, generator=g)
            b[:, cfg.x_dim + cfg.u_dim + 1] = 1.0
            b[:, D - 1] = (torch.rand(world * B, generator=g) < 0.2).float()
            nz = [torch.randn(world * B, cfg.u_dim, generator=g) for _ in range(3)]
            up.sgd_step(b[sl].to(dev), None, None, *[n[sl].to(dev) for n in nz])
            ref_rccl.sgd_step(b[sl].to(dev), None, None, *[n[sl].to(dev) for n in nz])
            ref.sgd_step(b.to(dev), None, None, *[n.to(dev) for n in nz])
            torch.cuda.synchronize()
            assert ex.status() == 0
            # identical to the collective path up to the summation order of two numbers (commutative: bit-exact for world 2)
            assert torch.equal(up.grads, ref_rccl.grads) and torch.equal(up.params, ref_rccl.params), it
            torch.testing.assert_close(up.params, ref.params, atol=5e-6, rtol=0)
        ex.close()
        (Path(tmpdir) / f"p2psac_ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("fused", [True, False])
def test_sac_sgd_step_over_peer_memory_two_ranks(tmp_path, fused):
    """SAC sgd_step with the gradient exchanged through peer memory on two ranks (half a minibatch each) — fused
    (mbpo_sac_grads_exchange_p2p -> mbpo_sac_apply: the reduction kernel stores, waits and sums) and split (mbpo_sac_grads_p2p ->
    mbpo_sac_gather_p2p -> mbpo_sac_apply) — equals the all-reduce path bit for bit and the single-process step on the whole
    minibatch within fp32 rounding."""
    world = 2
    port = 35500 + (os.getpid() % 2000) + (7 if fused else 0)
    mp.spawn(_p2p_sac_worker, args=(world, port, str(tmp_path), fused), nprocs=world, join=True)
    assert all((tmp_path / f"p2psac_ok{r}").exists() for r in range(world))


@pytest.mark.timeout(180)
def test_sac_layered_sgd_step_over_peer_memory_two_ranks(tmp_path):
    """The same exchange behind the layered forward/backward (hidden sizes outside the fused kernel's range: policy (96, 40), critics
    (200,)): its first launch moves the exchange's epoch words as block 0 of the fused kernel does."""
    world = 2
    port = 35500 + (os.getpid() % 2000) + 13
    mp.spawn(_p2p_sac_worker, args=(world, port, str(tmp_path), True, True), nprocs=world, join=True)
    assert all((tmp_path / f"p2psac_ok{r}").exists() for r in range(world))


def _bptt_setup(dev, pg, seed_key):
    from mbpo.optimizers import BPTTOptimizer
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.types import Transition
    system = PendulumSystem()
    s0 = system.reset()
    dummy = Transition(observation=s0.x_next, action=torch.zeros(1, device=dev), reward=s0.reward, discount=torch.tensor(0.99, device=dev),
                       next_observation=s0.x_next)
    q = UniformSamplingQueue(64, dummy, 1, device=dev)
    g = torch.Generator().manual_seed(0)
    th = (torch.rand(32, generator=g) * 2 - 1) * 3.14159
    obs = torch.stack([torch.cos(th), torch.sin(th), torch.zeros(32)], 1).to(dev)
    sbs = q.insert(q.init(0), Transition(observation=obs, action=torch.zeros(32, 1, device=dev), reward=torch.zeros(32, device=dev),
                                         discount=torch.ones(32, device=dev), next_observation=obs))
    opt = BPTTOptimizer(obs_dim=3, action_dim=1, horizon=6, num_samples_per_gradient_update=16, train_steps=5, sampling_buffer_size=4096,
                        process_group=pg)
    opt.set_system(system)
    return opt, opt.init(key=seed_key, true_buffer_state=sbs)


def _bptt_worker(rank, world, port, tmpdir, peer_exchange):
    """Two ranks on the one GPU.  peer_exchange: gradients and normaliser sums go through the peer-memory exchange and the train step
    IS captured (BPTT as SAC / PPO); otherwise the library collective — gloo here, host code — issued eagerly, never
    captured.  Either way the replicas stay bit-identical and the normaliser counts both ranks' transitions."""
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not peer_exchange:
        os.environ["MBPO_P2P_ALLREDUCE"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opt, st = _bptt_setup(dev, dist.group.WORLD, 100 + rank)     # different keys: the broadcast must make the nets identical
        assert (opt.p2p is not None) == peer_exchange and opt._capturable() == peer_exchange
        out = opt.train(bptt_state=st)
        assert opt._last_train_captured == peer_exchange
        o = out.optimizer_state
        for t in (o.actor_params, o.critic_params, o.target_critic_params, o.state_normalizer_state.vec):
            assert bool(torch.isfinite(t).all())
            ts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(ts, t)
            assert all(torch.equal(ts[0], x) for x in ts)            # replicas stay bit-identical
        assert float(o.state_normalizer_state.size) == 5 * 16 * 6 * world
        opt.close()
        (Path(tmpdir) / f"bptt_ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def _bptt_nccl_capture_worker(rank, world, port, tmpdir):
    """1-rank RCCL group: the BPTT train step holds torch.distributed (RCCL) all-reduces and IS captured; the result equals the
    group-less optimizer bit for bit (a 1-rank SUM is the identity)."""
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        opt, st = _bptt_setup(dev, dist.group.WORLD, 100)
        assert opt.p2p is None and opt._all_reduce is not None and opt._capturable()
        o = opt.train(bptt_state=st).optimizer_state
        assert opt._last_train_captured
        ref, st_ref = _bptt_setup(dev, None, 100)
        r = ref.train(bptt_state=st_ref).optimizer_state
        assert ref._last_train_captured
        for name in ("actor_params", "critic_params", "target_critic_params"):
            assert torch.equal(getattr(o, name), getattr(r, name)), name
        assert torch.equal(o.state_normalizer_state.vec, r.state_normalizer_state.vec)
        (Path(tmpdir) / "bptt_nccl_ok").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_sac_wide_critics_over_peer_memory_two_ranks(tmp_path):
    """wide layered critics (the reference's exp_ppo.py gives its critic (256,) x 5: NP ~ 530 k = 2070 reduction workgroups)
    are more than the fused in-kernel exchange may assume co-resident — the library refuses above 1024.  With the DEFAULT flags the
    updater takes the split exchange (push, gather, apply) for such networks instead of raising on the first update, and equals the
    all-reduce path bit for bit.  Run here with critics (256,) x 3 (NP ~ 276 k, 1077 workgroups, still past the limit): both ranks
    share this box's ONE GPU, and a waiting kernel of more workgroups than the device holds at once would leave no room for the
    peer's producer kernel until its bounded waits expire — an artefact of the rehearsal, not of separate GPUs."""
    world = 2
    port = 35500 + (os.getpid() % 2000) + 29
    mp.spawn(_p2p_sac_worker, args=(world, port, str(tmp_path), None, True, True), nprocs=world, join=True)
    assert all((tmp_path / f"p2psac_ok{r}").exists() for r in range(world))


@pytest.mark.timeout(240)
@pytest.mark.parametrize("peer_exchange", [True, False])
def test_bptt_data_parallel_two_ranks(tmp_path, peer_exchange):
    """BPTTOptimizer with a process group (bptt_optimizer.py:355-437 data-parallel): gradients and normaliser sums are reduced over the
    ranks — through the peer-memory exchange inside a captured train step, or through the library collective issued eagerly — and the
    replicas stay bit-identical."""
    world = 2
    port = 37500 + (os.getpid() % 2000) + (11 if peer_exchange else 0)
    mp.spawn(_bptt_worker, args=(world, port, str(tmp_path), peer_exchange), nprocs=world, join=True)
    assert all((tmp_path / f"bptt_ok{r}").exists() for r in range(world))


@pytest.mark.timeout(300)
def test_bptt_train_step_captured_with_rccl_collectives(tmp_path):
    """BPTT under an RCCL group captures its train step like SAC (tests above): 1-rank nccl captured == group-less, bit for bit."""
    port = 38500 + (os.getpid() % 2000)
    mp.spawn(_bptt_nccl_capture_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "bptt_nccl_ok").exists()


# ------------------------------------------------------------------------------------------------ trainer under a process group
def _trainer_setup(dev, pg, use_graph=True, n_steps=4):
    from mbpo.optimizers.policy_optimizers.sac.sac import SAC
    from mbpo.replay import UniformSamplingQueue
    from mbpo.systems import PendulumSystem
    from mbpo.systems.brax_wrapper import BraxWrapper
    from mbpo.types import Transition
    X, U, N, S = 3, 1, 64, 5
    system = PendulumSystem()
    dummy = Transition(observation=torch.zeros(X), action=torch.zeros(U), reward=torch.zeros(1), discount=torch.zeros(1),
                       next_observation=torch.zeros(X))
    tb = UniformSamplingQueue(128, dummy, 1, device=dev)
    g = torch.Generator().manual_seed(0)
    data = torch.randn(128, 2 * X + U + 2, generator=g)
    th = (torch.rand(128, generator=g) * 2 - 1) * 3.14159
    data[:, 0], data[:, 1] = torch.cos(th), torch.sin(th)
    tbs = tb.insert_rows(tb.init(0), data.to(dev))
    env = BraxWrapper(system, system.init_params(0), tbs, tb)
    tr = SAC(environment=env, num_timesteps=64 + N * S * n_steps, episode_length=5, num_env_steps_between_updates=S, num_envs=N,
             batch_size=64, grad_updates_per_step=3, normalize_observations=True, max_replay_size=1000, min_replay_size=64,
             use_graph=use_graph, process_group=pg)
    return tr, env


def _run_epoch(tr, env):
    rk = tr.dp.rank_key
    ts = tr.init_training_state(7)
    es = tr.reset_envs(env, rk(11), tr.num_envs)
    bs = tr.replay_buffer.init(rk(13))
    ts, es, bs, _ = tr.prefill_replay_buffer(ts, es, bs, rk(17))
    ts, es, bs, m = tr.training_epoch(ts, es, bs, rk(19))
    torch.cuda.synchronize()
    return ts, es, bs, m


def _nccl_capture_worker(rank, world, port, tmpdir):
    """1-rank RCCL group: the training step holds torch.distributed (RCCL) all-reduces and IS captured into a hipGraph
    (RCCL collectives are stream-ordered kernels).  Result == the group-less trainer, bit for bit (a 1-rank SUM is the identity)."""
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tr, env = _trainer_setup(dev, dist.group.WORLD)
        assert tr.p2p is None and tr._capturable()
        _run_epoch(tr, env)
        assert tr._graph is not None                              # captured WITH the collectives inside
        ref, env2 = _trainer_setup(dev, None)
        _run_epoch(ref, env2)
        for name in ("params", "target_q", "adam_m", "adam_v"):
            assert torch.equal(getattr(tr.updater, name), getattr(ref.updater, name)), name
        assert torch.equal(tr._stats_vec, ref._stats_vec)
        (Path(tmpdir) / "nccl_capture_ok").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sac_epoch_captured_with_rccl_collectives(tmp_path):
    """keep the hipGraph when the backend is nccl (RCCL).  In a child process: a capture that a collective
    invalidates cannot be recovered from in-process (round 1: hipErrorStreamCaptureInvalidated, then SIGSEGV) — here it must
    simply work."""
    port = 39500 + (os.getpid() % 2000)
    mp.spawn(_nccl_capture_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "nccl_capture_ok").exists()


def _trainer_dp_worker(rank, world, port, tmpdir, fail_rank):
    """Two ranks on the one GPU (gloo moves the tensors): per-rank DATA keys (rollout rows differ between ranks), identical
    parameters after the epoch, and — gloo being a host-side collective — the step is issued eagerly, never captured (the
    guarded path: round 1's abort was a gloo collective inside a capture).  fail_rank >= 0: that rank reports a failed peer-memory
    self-check; create() must decline on BOTH ranks without deadlock and training proceeds over the library collective."""
    _setup_paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if fail_rank >= 100:          # the peer exchange RAISES on that rank inside the timing stage: still no deadlock
        os.environ["MBPO_P2P_TEST_RAISE_TIMING_RANK"] = str(fail_rank - 100)
    elif fail_rank >= 0:
        os.environ["MBPO_P2P_TEST_FAIL_RANK"] = str(fail_rank)
    else:
        os.environ["MBPO_P2P_ALLREDUCE"] = "0"                    # library collective: the eager, uncaptured path
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr, env = _trainer_setup(dev, dist.group.WORLD)
        assert tr.p2p is None                                      # declined (forced failure) or switched off — on every rank
        from mbpo.parallel import P2PExchange
        assert P2PExchange.last_decline_reason                     # and every rank can say why (bench.py prints it)
        assert not tr._capturable()
        ts, es, bs, m = _run_epoch(tr, env)
        assert tr._graph is None                                   # never captured with a host-side collective inside
        rows = [torch.zeros_like(tr._rollout_rows) for _ in range(world)]
        dist.all_gather(rows, tr._rollout_rows)
        assert not torch.equal(rows[0], rows[1])                   # ranks roll out DIFFERENT envs with different noise
        for t in (tr.updater.params, tr.updater.target_q, tr._stats_vec):
            ts_ = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(ts_, t)
            assert torch.equal(ts_[0], ts_[1]) and bool(torch.isfinite(t).all())       # replicas stay bit-identical
        assert float(tr._stats_vec[0]) == world * (1 + 4) * 64 * 5                     # normaliser counts BOTH ranks' observations
        tr.close()
        (Path(tmpdir) / f"dp_ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fail_rank", [-1, 1, 101])
def test_sac_trainer_two_ranks_rank_keys_and_guarded_capture(tmp_path, fail_rank):
    world = 2
    port = 41500 + (os.getpid() % 2000) + (11 if fail_rank >= 0 else 0) + (7 if fail_rank >= 100 else 0)
    mp.spawn(_trainer_dp_worker, args=(world, port, str(tmp_path), fail_rank), nprocs=world, join=True)
    assert all((tmp_path / f"dp_ok{r}").exists() for r in range(world))


@pytest.mark.timeout(420)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as ONE command — no outer torch.distributed.run — starts its two ranks itself
    (the parent never touches the GPU), relays rank 0's single JSON line and exits 0.  Rehearsed with both ranks on the box's one
    GPU (MBPO_BENCH_SHARE_GPU=1: gloo process group — RCCL refuses two ranks on one device — peer-memory gradient exchange)."""
    import json
    import subprocess
    env = dict(os.environ, MBPO_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                                   # exactly one line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["pg_world_size"] == 2 and out["params_finite"]
    assert out["config"]["rank_device_ids"] == [0, 0]
    assert out["config"]["grad_exchange"].startswith("peer-memory") or out["config"]["p2p_decline_reason"]
    assert out["value"] > 0 and "steady_ms_per_step" in out
    # a rank that fails makes the whole command fail (here: more ranks than GPUs without the rehearsal switch)
    env.pop("MBPO_BENCH_SHARE_GPU")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
