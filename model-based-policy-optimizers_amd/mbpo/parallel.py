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
