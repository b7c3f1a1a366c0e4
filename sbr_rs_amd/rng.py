"""Host-side index RNG — mirror of the reference's ``rand::XorShiftRng`` usage
(``XorShiftRng::from_seed([42; 16])``, /root/reference/src/models/lstm.rs:428).

The algorithm is Marsaglia xorshift128 as specified in sbr_rs_amd/csrc/sbr_numerics.h
(``sbr_xorshift``); this class must stay bit-identical with it because a Python-side RNG is
handed to the engine as a 16-byte seed (``Hyperparameters.rng(rng)``, lstm.rs:122-125).
rand 0.5's exact streams are not pinned by any reference test (no rand source in this image), so
the streams are this engine's own.
"""
from __future__ import annotations

import numpy as np

_M32 = 0xFFFFFFFF
_M64 = (1 << 64) - 1


class XorShiftRng:
    def __init__(self, x: int, y: int, z: int, w: int):
        self.x, self.y, self.z, self.w = x & _M32, y & _M32, z & _M32, w & _M32

    @classmethod
    def from_seed(cls, seed) -> "XorShiftRng":
        seed = bytes(seed)
        if len(seed) != 16:
            raise ValueError("seed must be 16 bytes")
        s = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(4)]
        if not any(s):
            s = [0x193A6754, 0xA8A7D469, 0x97830E05, 0x113BA7BB]
        return cls(*s)

    def state_seed(self) -> bytes:
        """The 16 bytes that re-create the current state through ``from_seed``."""
        return b"".join(int(v).to_bytes(4, "little") for v in (self.x, self.y, self.z, self.w))

    def clone(self) -> "XorShiftRng":
        return XorShiftRng(self.x, self.y, self.z, self.w)

    def next_u32(self) -> int:
        t = (self.x ^ (self.x << 11)) & _M32
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & _M32
        return self.w

    def next_u64(self) -> int:
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)

    def below(self, n: int) -> int:
        """Uniform integer in [0, n): 64x64->128 multiply-high with rejection (sbr_xs_below)."""
        thresh = ((1 << 64) - n) % n
        while True:
            m = self.next_u64() * n
            if (m & _M64) >= thresh:
                return m >> 64

    def unit(self) -> float:
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def permutation(self, n: int) -> np.ndarray:
        """Fisher-Yates from the end: ``for i in (1..n).rev(): swap(i, below(i+1))``."""
        perm = np.arange(n, dtype=np.int64)
        for i in range(n, 1, -1):
            j = self.below(i)
            perm[i - 1], perm[j] = perm[j], perm[i - 1]
        return perm
