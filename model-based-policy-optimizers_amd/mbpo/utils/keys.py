"""Explicit PRNG keys for the host API.

The reference threads jax.random.PRNGKey arrays through every call and splits them (e.g. brax_optimizers.py:81-83,
sac/sac.py:405-409).  JAX's threefry streams cannot be reproduced without JAX, so a key here is a 64-bit integer and
`split` derives children with splitmix64; the SAME call structure as the reference is kept (who splits what, when).
Device kernels turn (key, counter, stream id, element index) into numbers with Philox4x32-10 (csrc/common.hpp).
"""
from __future__ import annotations

from typing import List, Union

MASK = (1 << 64) - 1
Key = int


def PRNGKey(seed: Union[int, "Key"]) -> Key:
    return int(seed) & MASK


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def split(key: Key, num: int = 2) -> List[Key]:
    """jax.random.split analogue: `num` statistically independent child keys."""
    base = _splitmix64(int(key) & MASK)
    return [_splitmix64((base + i * 0xD1B54A32D192ED03) & MASK) for i in range(num)]
