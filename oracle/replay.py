"""UniformSamplingQueue and running_statistics — numpy restatement (test infrastructure).

[3P, unverifiable here] brax.training.replay_buffers.UniformSamplingQueue / brax.training.acme.running_statistics
are not in the reference tree; call sites: sac/sac.py:202-205,303,318,326,415; systems/brax_wrapper.py:29;
base_optimizer.py:54-57; bptt_optimizer.py:258-261,447-456,459,479.  State field use (data, insert_position,
sample_position, key) is evidenced by bptt_optimizer.py:447-455.

This restatement keeps the reference's LOGICAL array and really rolls it (np.roll), so it is an independent
check of the product's O(1) ring-offset implementation.
"""
from __future__ import annotations

import numpy as np

from . import philox


class UniformSamplingQueue:
    def __init__(self, max_replay_size: int, row_len: int, sample_batch_size: int):
        self.max = int(max_replay_size)
        self.D = int(row_len)
        self.sample_batch_size = int(sample_batch_size)

    def init(self):
        """QueueBase.init: zeros data, int32 zero positions."""
        return {"data": np.zeros((self.max, self.D), np.float32), "insert_position": np.int32(0),
                "sample_position": np.int32(0)}

    def insert(self, st, rows: np.ndarray):
        """QueueBase.insert_internal."""
        rows = np.asarray(rows, np.float32)
        n = rows.shape[0]
        if n > self.max:
            raise ValueError("update larger than the buffer")
        data = st["data"]
        position = int(st["insert_position"])
        roll = min(0, self.max - position - n)
        if roll:
            data = np.roll(data, roll, axis=0)
        else:
            data = data.copy()
        position = position + roll
        data[position:position + n] = rows                      # dynamic_update_slice_in_dim
        position = (position + n) % (self.max + 1)
        sample_position = max(0, int(st["sample_position"]) + roll)
        return {"data": data, "insert_position": np.int32(position), "sample_position": np.int32(sample_position)}

    def size(self, st) -> int:
        return int(st["insert_position"]) - int(st["sample_position"])

    def sample_indices(self, st, seed: int, offset: int, n: int | None = None) -> np.ndarray:
        """jax.random.randint(key, (batch,), sample_position, insert_position) restated with the build's Philox."""
        n = self.sample_batch_size if n is None else n
        return philox.philox_randint(seed, offset, philox.STREAM_REPLAY, np.arange(n, dtype=np.uint64),
                                     int(st["sample_position"]), int(st["insert_position"]))

    def gather(self, st, idx: np.ndarray) -> np.ndarray:
        """jnp.take(data, idx, axis=0, mode='wrap')."""
        return st["data"][np.mod(np.asarray(idx, np.int64), self.max)]

    def sample(self, st, seed: int, offset: int, n: int | None = None):
        idx = self.sample_indices(st, seed, offset, n)
        return idx, self.gather(st, idx)


# ---------------------------------------------------------------------------------------- running statistics
def stats_init(x_dim: int) -> np.ndarray:
    """running_statistics.init_state: count 0, mean 0, summed_variance 0, std 1.  Layout [count, mean, sv, std]."""
    s = np.zeros(1 + 3 * x_dim, np.float32)
    s[1 + 2 * x_dim:] = 1.0
    return s


def stats_update(stats: np.ndarray, batch: np.ndarray, std_min=1e-6, std_max=1e6, dtype=np.float32) -> np.ndarray:
    """running_statistics.update (two-pass form exactly as in acme):
        count += n; diff_to_old = batch - mean; mean += sum(diff_to_old)/count;
        diff_to_new = batch - mean; summed_variance += sum(diff_to_old*diff_to_new);
        std = clip(sqrt(max(summed_variance,0)/count), std_min, std_max)
    """
    x = batch.shape[1]
    s = stats.astype(dtype)
    batch = batch.astype(dtype)
    count = s[0] + dtype(batch.shape[0])
    mean, sv = s[1:1 + x], s[1 + x:1 + 2 * x]
    diff_old = batch - mean
    mean_new = mean + diff_old.sum(axis=0) / count
    diff_new = batch - mean_new
    sv_new = sv + (diff_old * diff_new).sum(axis=0)
    std = np.clip(np.sqrt(np.maximum(sv_new, 0) / count), std_min, std_max)
    return np.concatenate([[count], mean_new, sv_new, std]).astype(dtype)
