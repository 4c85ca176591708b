"""Philox4x32-10 counter-based generator — numpy restatement of csrc/common.hpp (build-defined, not in the
reference: JAX's threefry streams cannot be reproduced here).  Integer parts are bit-exact.
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)

MASK64 = (1 << 64) - 1

STREAM_POLICY_NOISE = 1
STREAM_MODEL_NOISE = 2
STREAM_MEMBER = 3
STREAM_REPLAY = 4
STREAM_SAC_ALPHA = 5
STREAM_SAC_CRITIC = 6
STREAM_SAC_ACTOR = 7
STREAM_PERM = 8
STREAM_ENTROPY = 9
STREAM_ICEM = 10


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays. Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape).copy()
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n2 = hi0 ^ c3 ^ k1
            c0, c1, c2, c3 = n0, lo1, n2, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _counters(seed: int, offset: int, stream: int, idx: np.ndarray):
    idx = np.asarray(idx, dtype=np.uint64)
    c0 = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (idx >> np.uint64(32)).astype(np.uint32)
    c2 = np.uint32((stream ^ (((offset >> 32) & 0xFFFFFFFF) * 0x9E3779B9)) & 0xFFFFFFFF)
    c3 = np.uint32(offset & 0xFFFFFFFF)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    return c0, c1, c2, c3, k0, k1


def philox_normal(seed: int, offset: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """float32 standard normal for each element index (Box-Muller on words 0,1)."""
    r0, r1, _, _ = philox4x32_10(*_counters(seed, offset, stream, idx))
    u1 = ((r0 >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    u2 = (r1 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
    return (rad * np.cos(np.float32(6.28318530717958647692) * u2, dtype=np.float32)).astype(np.float32)


def philox_randint(seed: int, offset: int, stream: int, idx: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """int32 uniform in [lo, hi): lo + mulhi(u32, hi-lo).  Bit-exact with the device."""
    r0, _, _, _ = philox4x32_10(*_counters(seed, offset, stream, idx))
    span = np.uint64(hi - lo)
    return (np.int64(lo) + ((r0.astype(np.uint64) * span) >> np.uint64(32)).astype(np.int64)).astype(np.int32)


def resolve(seed: int, offset: int, rng=None):
    """csrc/common.hpp:rng_resolve — the device RNG words {seed word, step counter} are ADDED to the host (seed, offset), mod 2^64."""
    if rng is None:
        return seed & MASK64, offset & MASK64
    return (seed + int(rng[0])) & MASK64, (offset + int(rng[1])) & MASK64


def philox_u32(seed: int, offset: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """Word 0 of the Philox block of each element index."""
    r0, _, _, _ = philox4x32_10(*_counters(seed, offset, stream, idx))
    return r0


def philox_permutation(seed: int, offset: int, n: int) -> np.ndarray:
    """csrc/replay.hip:mbpo_philox_permutation — stable argsort of the PERM-stream keys."""
    keys = philox_u32(seed, offset, STREAM_PERM, np.arange(n, dtype=np.uint64))
    return np.argsort(keys, kind="stable").astype(np.int32)


# ---------------------------------------------------------------------------------------------- host keys
def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def split(key: int, num: int = 2):
    """Restatement of the product's host key derivation (mbpo/utils/keys.py: build-defined — JAX's threefry split cannot be
    reproduced here): `num` child keys by splitmix64."""
    base = _splitmix64(int(key) & MASK64)
    return [_splitmix64((base + i * 0xD1B54A32D192ED03) & MASK64) for i in range(num)]
