"""world_size-2 gloo tests on CPU: the data-parallel semantics of SURVEY §8e with the product's host glue
(mbpo.parallel.DataParallel) and the oracle as the arithmetic.

  * pmean of per-rank minibatch gradients == gradient of the global minibatch (all three SAC losses are means);
    clip must see the POST-reduce norm (sac/utils.py:57-60);
  * running_statistics.update with a psum after each pass == the update on the concatenated batch;
  * rank keys differ, parameter broadcast makes replicas identical.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, tmpdir):
    for p in (str(ROOT), str(ROOT / "model-based-policy-optimizers_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mbpo.parallel import DataParallel
        from oracle import replay as orep
        from oracle import sac as osac
        torch.set_num_threads(1)
        dp = DataParallel(dist.group.WORLD)
        assert dp.world_size == world and dp.rank == rank
        assert list(dp.shard(8)) == list(range(rank * 4, rank * 4 + 4))
        keys = [None] * world
        dist.all_gather_object(keys, dp.rank_key(123))
        assert len(set(keys)) == world                      # per-rank keys differ
        # no GPU here: the peer-memory exchange declines on every rank and the collective path stays
        from mbpo.parallel import P2PExchange
        assert P2PExchange.create(dp, 1000, "cuda:0") is None

        # ---- parameter broadcast
        g = torch.Generator().manual_seed(100 + rank)        # deliberately different per rank
        X, U, B = 4, 1, 32
        cfg = osac.SacConfig(X, U, [X, 64, 64, 2 * U], [X + U, 64, 64, 1], max_grad_norm=0.05, lr_policy=1e-3, lr_q=1e-3,
                             lr_alpha=1e-3)
        st = osac.init_state(cfg, g)
        dp.broadcast(st.params)
        st.target_q = st.params[cfg.P:cfg.P + 2 * cfg.Q].clone()
        gathered = [torch.zeros_like(st.params) for _ in range(world)]
        dist.all_gather(gathered, st.params)
        assert all(torch.equal(gathered[0], t) for t in gathered)

        # ---- pmean(grad) == grad of the global batch; clip after the reduce
        gg = torch.Generator().manual_seed(7)                # same on every rank: the GLOBAL batch
        D = 2 * X + U + 3
        batch = torch.randn(world * B, D, generator=gg)
        batch[:, X + U + 1] = 1.0
        batch[:, D - 1] = (torch.rand(world * B, generator=gg) < 0.2).float()
        noise = [torch.randn(world * B, U, generator=gg) for _ in range(3)]
        sl = slice(rank * B, (rank + 1) * B)
        g_local, _ = osac.grads(cfg, st.params, st.target_q, batch[sl], *[n[sl] for n in noise])
        g_sum = dp.all_reduce_sum(g_local.clone())
        g_mean = g_sum / world
        g_full, _ = osac.grads(cfg, st.params, st.target_q, batch, *noise)
        torch.testing.assert_close(g_mean, g_full, atol=1e-6, rtol=1e-5)
        assert float(torch.sqrt((g_full[:cfg.P] ** 2).sum())) > cfg.max_grad_norm      # the clip really triggers
        st_dp, _, _ = osac.sgd_step(cfg, st, batch[sl], *[n[sl] for n in noise], grad_override=g_mean)
        st_full, _, _ = osac.sgd_step(cfg, st, batch, *noise)
        torch.testing.assert_close(st_dp.params, st_full.params, atol=2e-6, rtol=0)
        # clipping each rank's LOCAL gradient before the reduce would be a different (wrong) update
        g_wrong = dp.all_reduce_sum(osac.clip_by_global_norm(g_local[:cfg.P], cfg.max_grad_norm).clone()) / world
        assert not torch.allclose(g_wrong, osac.clip_by_global_norm(g_full[:cfg.P], cfg.max_grad_norm), atol=1e-6)

        # ---- running statistics: psum after each pass
        rng = np.random.default_rng(3)
        obs = (rng.standard_normal((world * 50, X)) * 2 + 1).astype(np.float64)
        mine = obs[rank * 50:(rank + 1) * 50]
        stats = orep.stats_init(X).astype(np.float64)
        stats[0], stats[1:1 + X] = 10.0, 0.3                 # a non-trivial previous state
        mean_old = stats[1:1 + X]
        sums = torch.zeros(1 + 2 * X, dtype=torch.float64)
        sums[0] = mine.shape[0]
        sums[1:1 + X] = torch.from_numpy((mine - mean_old).sum(0))
        dp.all_reduce_sum(sums)                              # psum #1
        upd = sums[1:1 + X].numpy() / (stats[0] + float(sums[0]))
        d = mine - mean_old
        sums[1 + X:] = torch.from_numpy((d * (d - upd)).sum(0))
        dp.all_reduce_sum(sums[1 + X:])                      # psum #2
        count = stats[0] + float(sums[0])
        mean_new = mean_old + sums[1:1 + X].numpy() / count
        sv_new = stats[1 + X:1 + 2 * X] + sums[1 + X:].numpy()
        ref = orep.stats_update(stats, obs, dtype=np.float64)
        np.testing.assert_allclose(np.concatenate([[count], mean_new, sv_new]), ref[:1 + 2 * X], rtol=1e-12)
        # ---- staged agreement (P2PExchange.create): rank 1 fails stage 1; stage 2 holds a collective and must then run on
        #      NO rank (if rank 0 entered it alone it would pair its all_reduce with rank 1's next collective and hang)
        from mbpo.parallel import run_agreed_stages
        ran = []

        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag) == 1

        def stage1():
            ran.append(1)
            return rank != 1

        def stage2():
            ran.append(2)
            dist.all_reduce(torch.zeros(1))
            return True

        assert run_agreed_stages([stage1, stage2], agree) is False and ran == [1]
        ran.clear()

        def stage_raises():
            ran.append(1)
            if rank == 0:
                raise RuntimeError("local failure")
            return True

        assert run_agreed_stages([stage_raises, stage2], agree) is False and ran == [1]
        ran.clear()
        assert run_agreed_stages([lambda: True, stage2], agree) is True and ran == [2]
        marker = torch.tensor([float(rank)])
        dist.all_reduce(marker)                              # the ranks' collective sequences are still aligned
        assert float(marker) == sum(range(world))
        (Path(tmpdir) / f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_data_parallel_semantics_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_data_parallel_single_process_noop():
    sys.path.insert(0, str(ROOT / "model-based-policy-optimizers_amd"))
    from mbpo.parallel import DataParallel
    dp = DataParallel(None)
    t = torch.arange(4.0)
    assert dp.all_reduce_fn() is None and torch.equal(dp.all_reduce_sum(t.clone()), t) and dp.rank_key(5) == 5
    assert list(dp.shard(6)) == list(range(6))
