"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference threads `pmap_axis_name` through its update functions but hard-codes `_PMAP_AXIS_NAME = None`
(sac/sac.py:188-189, ppo/ppo.py:96-97), so its one collective, `jax.lax.pmean(grad)` (sac/utils.py:29-33), and the
`psum`s inside `running_statistics.update(..., pmap_axis_name)` (sac/sac.py:298-301) are dead code.  This class is their
live form: parallel envs, the model-replay shard and the minibatch are per rank; the ONLY exchanges are
  * one SUM all-reduce of the flat gradient per sgd_step (scaled by 1/world inside mbpo_sac_apply -> pmean),
  * two SUM all-reduces of the normaliser's sufficient statistics per training_step,
  * a broadcast of rank 0's initial parameters.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from mbpo.utils import keys as K


class DataParallel:
    def __init__(self, process_group=None):
        self.group = process_group
        if process_group is None:
            self.world_size, self.rank = 1, 0
        else:
            import torch.distributed as dist
            self.world_size = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)

    @property
    def active(self) -> bool:
        return self.world_size > 1 or self.group is not None

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM over ranks (no-op without a group)."""
        if self.group is not None:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_fn(self):
        """Callable for ops.SacUpdater / ops.running_stats_update, or None when there is nothing to reduce."""
        return self.all_reduce_sum if self.group is not None else None

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.group is not None:
            import torch.distributed as dist
            dist.broadcast(t, src=src, group=self.group)
        return t

    def rank_key(self, key: int) -> int:
        """Per-rank key for everything that must DIFFER across ranks (env resets, rollout noise, replay sampling)."""
        return K.split(key, self.world_size)[self.rank] if self.world_size > 1 else key

    def shard(self, n: int) -> range:
        """Contiguous shard of `n` units (envs) owned by this rank; n must divide evenly."""
        if n % self.world_size:
            raise ValueError(f"{n} units do not shard evenly over {self.world_size} ranks")
        per = n // self.world_size
        return range(self.rank * per, (self.rank + 1) * per)


def run_agreed_stages(stages, agree) -> bool:
    """Run `stages` (callables returning a rank-local ok flag; they may contain collectives) one after the other, with an
    agreement after EVERY stage: `agree(ok)` is a collective that returns True only if every rank's ok was True.  A rank whose
    stage fails (returns False or raises) still takes part in that stage's agreement and no rank starts the next stage unless
    all ranks passed this one — so the ranks' collective sequences can never diverge (a rank-local early return
    before a stage that contains collectives deadlocks the other ranks).  Returns True iff every stage passed everywhere."""
    for stage in stages:
        try:
            ok = bool(stage())
        except Exception:      # noqa: BLE001 — any local failure means "not ok", never a skipped collective
            ok = False
        if not agree(ok):
            return False
    return True


class P2PExchange:
    """Peer-memory exchange regions for the one-shot all-reduce (csrc/p2p.hpp): every rank allocates a region, the 64-byte IPC
    handles travel through the process group (all_gather_object), every rank maps every peer's region.

    `create` returns None (the caller then keeps the RCCL all-reduce) when the regions cannot be set up, when
    MBPO_P2P_ALLREDUCE=0, or when the self-check — one exchange of a rank-dependent vector compared with
    torch.distributed.all_reduce — does not reproduce the library's result on every rank."""

    # why the most recent create() returned None on this rank (bench.py prints it; None after a successful create)
    last_decline_reason: Optional[str] = None

    def __init__(self, dp: "DataParallel", n_max: int, device: torch.device):
        from mbpo import _hip
        self.lib = _hip.load()
        self._hip = _hip
        self.dp, self.n_max, self.device = dp, int(n_max), device
        self.own = C.c_void_p()
        self.peers = {}
        self.desc = None

    @classmethod
    def create(cls, dp: "DataParallel", n_max: int, device) -> Optional["P2PExchange"]:
        """Collective over dp.group.  Every step that can fail locally is followed by an agreement (MIN all-reduce of an ok
        flag) before the next collective, so one rank's failure turns into `None` on every rank instead of a deadlock."""
        cls.last_decline_reason = None
        if dp.group is None or dp.world_size < 2:
            cls.last_decline_reason = "single rank: nothing to exchange"
            return None
        if os.environ.get("MBPO_P2P_ALLREDUCE", "1") == "0":
            cls.last_decline_reason = "disabled by MBPO_P2P_ALLREDUCE=0"
            return None
        if not torch.cuda.is_available():       # no device, no peer memory: the (gloo) collective stays
            cls.last_decline_reason = "no GPU visible"
            return None
        import torch.distributed as dist
        device = torch.device(device)
        ex = cls(dp, n_max, device)

        def agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=dp.group)
            return int(flag) == 1

        # 1. own region + IPC handle (local)
        handle = None
        try:
            handle = ex._alloc()
        except Exception as e:      # noqa: BLE001 — any failure means: use the library collective
            ex._err = repr(e)
        if not agree(handle is not None):
            cls.last_decline_reason = f"region allocation / IPC handle failed on some rank (this rank: {getattr(ex, '_err', 'ok')})"
            ex.close()
            return None
        # 2. exchange the handles (collective), map the peers (local)
        mine = (handle, torch.cuda.current_device() if device.index is None else device.index, os.getpid())
        gathered = [None] * dp.world_size
        dist.all_gather_object(gathered, mine, group=dp.group)
        ok = True
        try:
            ex._open_peers(gathered, mine[1])
        except Exception as e:      # noqa: BLE001
            ok = False
            ex._err = repr(e)
        if not agree(ok):
            cls.last_decline_reason = f"mapping a peer's region failed on some rank (this rank: {getattr(ex, '_err', 'ok')})"
            ex.close()
            return None
        # 3. three exchanges against the library all-reduce, then — only if EVERY rank passed them — the timing of both
        #    (the timing stage contains collectives of its own: it must not start on some ranks only)
        if not run_agreed_stages([ex._check_correctness, ex._check_timing], agree):
            tm = getattr(ex, "timing_ms", None)
            cls.last_decline_reason = (
                f"start-up check failed on some rank (this rank: status {ex._safe_status()}, "
                + (f"exchange {tm[0]:.3f} ms vs library all-reduce {tm[1]:.3f} ms" if tm else "correctness stage")
                + f", error {getattr(ex, '_err', None)})")
            ex.close()
            return None
        return ex

    def _alloc(self) -> bytes:
        _hip, lib, dp = self._hip, self.lib, self.dp
        nbytes = lib.mbpo_p2p_region_bytes(dp.world_size, self.n_max)
        if nbytes < 0:
            _hip.check(int(nbytes), "mbpo_p2p_region_bytes")
        handle = (C.c_ubyte * 64)()
        _hip.check(lib.mbpo_p2p_alloc(nbytes, C.byref(self.own), handle), "mbpo_p2p_alloc")
        return bytes(handle)

    def _open_peers(self, gathered, my_dev: int):
        _hip, lib, dp = self._hip, self.lib, self.dp
        d = _hip.P2pDesc()
        d.world, d.rank, d.n_max = dp.world_size, dp.rank, self.n_max
        for r, (h, dev_idx, _pid) in enumerate(gathered):
            if r == dp.rank:
                d.regions[r] = self.own.value
                continue
            p = C.c_void_p()
            hb = (C.c_ubyte * 64).from_buffer_copy(h)
            _hip.check(lib.mbpo_p2p_open(hb, -1 if dev_idx == my_dev else dev_idx, C.byref(p)), "mbpo_p2p_open")
            self.peers[r] = p
            d.regions[r] = p.value
        self.desc = d

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM of a contiguous fp32 device tensor (numel <= n_max) over the ranks."""
        _hip = self._hip
        _hip.require_device_tensor(t, "t")
        if t.numel() > self.n_max:
            raise ValueError(f"tensor has {t.numel()} elements, the exchange regions hold {self.n_max}")
        _hip.check(self.lib.mbpo_p2p_all_reduce_sum(C.byref(self.desc), t.data_ptr(), t.numel(), _hip.current_stream_ptr()),
                   "mbpo_p2p_all_reduce_sum")
        return t

    def status(self) -> int:
        v = C.c_int32(0)
        self._hip.check(self.lib.mbpo_p2p_status(C.byref(self.desc), C.byref(v)), "mbpo_p2p_status")
        return int(v.value)

    def _safe_status(self):
        try:
            return self.status()
        except Exception:      # noqa: BLE001
            return "unreadable"

    def _check_correctness(self) -> bool:
        import torch.distributed as dist
        n = min(self.n_max, 4099)
        ok = True
        for it in range(3):      # three exchanges: both slot parities and a re-use; every rank runs all of them
            g = torch.Generator().manual_seed(1000 * it + self.dp.rank)
            x = torch.randn(n, generator=g).to(self.device)
            ref = x.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=self.dp.group)      # issued on every rank whatever happened before
            try:
                self.all_reduce_sum(x)       # bounded waits inside: a missing peer ends in a status flag, not a hang
                torch.cuda.synchronize()
                ok = ok and self.status() == 0 and bool(torch.isfinite(x).all()) and torch.allclose(x, ref, rtol=1e-5, atol=1e-5)
            except Exception as e:      # noqa: BLE001
                ok = False
                self._err = repr(e)
        if os.environ.get("MBPO_P2P_TEST_FAIL_RANK") == str(self.dp.rank):     # test hook: this rank reports a failed check
            ok = False
        return ok

    def _check_timing(self) -> bool:
        """A mapping that works but is pathologically slow (e.g. peer stores routed over PCIe) must not replace the library
        collective: time both on a gradient-sized vector; every rank runs the same loops (the exchange is a collective)."""
        import torch.distributed as dist
        n = self.n_max
        x = torch.zeros(n, device=self.device)
        y = torch.zeros(n, device=self.device)
        reps = 10

        local_ok = [True]

        def guarded_exchange():
            # only the peer exchange may fail on one rank alone (HIP error, bad status): it is recorded, never raised — the
            # library collectives of this stage (the barriers, the all-reduce loop) are issued by EVERY rank whatever happened,
            # so a local failure cannot leave the peers in mismatched collectives
            try:
                if os.environ.get("MBPO_P2P_TEST_RAISE_TIMING_RANK") == str(self.dp.rank):     # test hook
                    raise RuntimeError("forced failure of the peer exchange in the timing stage")
                self.all_reduce_sum(x)
            except Exception as e:      # noqa: BLE001
                local_ok[0] = False
                self._err = repr(e)

        def timed(fn):
            fn()
            dist.barrier(group=self.dp.group)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        # one exchange first, then an agreement: if it failed on ANY rank nobody enters the timed loop (a rank that has given up
        # would leave its peers in one bounded wait of 20-30 s per repetition)
        guarded_exchange()
        torch.cuda.synchronize()
        flag = torch.tensor([1 if local_ok[0] and self._safe_status() == 0 else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.dp.group)
        if int(flag) != 1:
            return False
        t_p2p = timed(guarded_exchange)
        t_lib = timed(lambda: dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.dp.group))
        self.timing_ms = (t_p2p, t_lib)
        return local_ok[0] and self.status() == 0 and t_p2p <= 4.0 * t_lib + 0.05

    def close(self):
        for p in self.peers.values():
            try:
                self.lib.mbpo_p2p_close(p)
            except Exception:       # noqa: BLE001
                pass
        self.peers = {}
        if self.own.value:
            try:
                self.lib.mbpo_p2p_free(self.own)
            except Exception:       # noqa: BLE001
                pass
            self.own = C.c_void_p()
