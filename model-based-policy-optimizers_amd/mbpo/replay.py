"""UniformSamplingQueue — device-resident replay buffer with the brax semantics the reference relies on
(call sites: sac/sac.py:202-205,303,318,326,415; systems/brax_wrapper.py:29; base_optimizer.py:54-57;
bptt_optimizer.py:258-261,447-456).  Insert / sample / gather run in libmbpo_hip.so (csrc/replay.hip).

Differences a reference user should know:
  * the reference is functional (insert returns a fresh array); here `data` is mutated in place and the returned
    ReplayBufferState aliases it — keep only the newest state, as the reference's call sites do anyway;
  * positions live on the device (graph-replayable) AND are mirrored on the host as Python ints (same integer
    arithmetic; `size()` needs no sync);
  * `key` is an integer (mbpo.utils.keys); sampling draws Philox numbers on the device.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import ops
from .types import Transition, flatten, row_layout, unflatten
from .utils import keys as K


@dataclass
class ReplayBufferState:
    data: torch.Tensor            # [max_replay_size, D] physical ring storage
    state: torch.Tensor           # device int32[4]: insert_position, sample_position, head, total_inserted
    key: int
    insert_position: int = 0      # host mirrors
    sample_position: int = 0
    head: int = 0
    sample_count: int = 0         # number of sample() calls so far (Philox offset)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class UniformSamplingQueue:
    def __init__(self, max_replay_size: int, dummy_data_sample: Transition, sample_batch_size: int, device=None):
        self.max_replay_size = int(max_replay_size)
        self.sample_batch_size = int(sample_batch_size)
        d = dummy_data_sample
        self.x_dim = int(d.observation.reshape(-1).shape[0])
        self.u_dim = int(d.action.reshape(-1).shape[0])
        ex = d.extras or {}
        self.has_truncation = "state_extras" in ex and "truncation" in ex["state_extras"]
        self.ppo_extras = "policy_extras" in ex and "log_prob" in ex["policy_extras"]
        _, self.row_len = row_layout(self.x_dim, self.u_dim, self.has_truncation, self.ppo_extras)
        self.device = torch.device(device) if device is not None else (
            d.observation.device if d.observation.is_cuda else torch.device("cuda", torch.cuda.current_device()))

    # -- reference API ------------------------------------------------------------------------------------------
    def init(self, key: int) -> ReplayBufferState:
        data = torch.zeros(self.max_replay_size, self.row_len, device=self.device, dtype=torch.float32)
        state = torch.zeros(4, device=self.device, dtype=torch.int32)
        return ReplayBufferState(data=data, state=state, key=K.PRNGKey(key))

    def insert(self, buffer_state: ReplayBufferState, samples) -> ReplayBufferState:
        rows = samples if isinstance(samples, torch.Tensor) else flatten(samples)
        return self.insert_rows(buffer_state, rows)

    def sample(self, buffer_state: ReplayBufferState) -> Tuple[ReplayBufferState, Transition]:
        st, rows = self.sample_rows(buffer_state)
        return st, unflatten(rows, self.x_dim, self.u_dim, self.has_truncation, self.ppo_extras)

    def size(self, buffer_state: ReplayBufferState) -> int:
        return buffer_state.insert_position - buffer_state.sample_position

    # -- row-level fast path (what the trainers use) --------------------------------------------------------------
    def insert_rows(self, bs: ReplayBufferState, rows: torch.Tensor) -> ReplayBufferState:
        n = rows.shape[0]
        if n > self.max_replay_size:
            raise ValueError(f"cannot insert {n} rows into a buffer of {self.max_replay_size}")
        ops.replay_insert(bs.data, bs.state, rows.to(self.device).contiguous())
        return self.insert_mirror(bs, n)

    def insert_mirror(self, bs: ReplayBufferState, n: int) -> ReplayBufferState:
        """Host mirror of one insert of n rows — same integer arithmetic as csrc/replay.hip:replay_plan / k_replay_advance
        (also used alone when the insert itself ran inside a replayed hipGraph)."""
        mx = self.max_replay_size
        roll = min(0, mx - bs.insert_position - n)
        pos = bs.insert_position + roll
        return bs.replace(insert_position=(pos + n) % (mx + 1), sample_position=max(0, bs.sample_position + roll),
                          head=(bs.head - roll) % mx)

    def sample_rows(self, bs: ReplayBufferState, n: Optional[int] = None, out: Optional[torch.Tensor] = None,
                    rng_dev: Optional[torch.Tensor] = None, seed: Optional[int] = None, offset: int = 0
                    ) -> Tuple[ReplayBufferState, torch.Tensor]:
        """`seed` overrides the key-derived sample seed (the trainers pass 0 and key the draw through `rng_dev`, so that a
        captured hipGraph and the eager path draw the same indices)."""
        n = self.sample_batch_size if n is None else n
        if self.size(bs) <= 0:
            raise ValueError("cannot sample from an empty replay buffer")
        key, sample_key = K.split(bs.key)          # QueueBase.sample_internal: key, sample_key = split(key)
        rows = ops.replay_sample(bs.data, bs.state, n, seed=sample_key if seed is None else seed, offset=offset, out=out,
                                 rng_dev=rng_dev)
        return bs.replace(key=key, sample_count=bs.sample_count + 1), rows

    def logical_data(self, bs: ReplayBufferState) -> torch.Tensor:
        """The reference's `data` array (logical row order), materialised — for checkpoints/tests, not the hot path."""
        idx = (torch.arange(self.max_replay_size, device=bs.data.device) + bs.head) % self.max_replay_size
        return bs.data[idx]
