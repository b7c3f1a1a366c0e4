"""Host-side index RNG — mirror of the reference's ``rand::XorShiftRng`` usage
(``XorShiftRng::from_seed([42; 16])``, /root/reference/src/models/lstm.rs:428).

rand 0.5 as recalled (the crate's source is not in this image and no reference test
pins a stream): Marsaglia xorshift128, ``next_u64`` = low word first, ``gen_range`` / ``shuffle`` =
``UniformInt::sample_single`` (zone = range << leading_zeros), ``Uniform::new`` = modulus zone,
``gen::<[u8; 16]>`` = sixteen truncated ``next_u32``.  Must stay bit-identical with ``sbr_xorshift`` /
``sbr_rand_*`` in sbr_rs_amd/csrc/sbr_numerics.h: a Python-side RNG is handed to the engine as a
16-byte seed (``Hyperparameters.rng(rng)``, lstm.rs:122-125).
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

    def gen_range(self, low: int, high: int) -> int:
        """``Rng::gen_range(low, high)`` = ``UniformInt::sample_single``: zone = range << leading_zeros(range);
        draw v = next_u64, accept when the low half of v * range is <= zone; result = low + high half.  The crate
        asserts low < high; here: ValueError."""
        if low >= high:
            raise ValueError("gen_range: low must be below high")
        rng = high - low
        zone = (rng << (64 - rng.bit_length())) & _M64
        while True:
            m = self.next_u64() * rng
            if (m & _M64) <= zone:
                return low + (m >> 64)

    def uniform(self, low: int, high: int) -> int:
        """``Uniform::new(low, high).sample(rng)``: zone = MAX - (MAX - range + 1) % range (data.rs:77-78)."""
        if low >= high:
            raise ValueError("Uniform::new: low must be below high")
        rng = high - low
        zone = _M64 - (_M64 - rng + 1) % rng
        while True:
            m = self.next_u64() * rng
            if (m & _M64) <= zone:
                return low + (m >> 64)

    def below(self, n: int) -> int:
        """``gen_range(0, n)``."""
        return self.gen_range(0, n)

    def gen_seed(self) -> bytes:
        """``rng.gen::<[u8; 16]>()``: sixteen ``next_u32`` calls, each truncated to its low byte
        (``XorShiftRng::from_seed(parameters.rng().gen())``, sequence_model.rs:97)."""
        return bytes(self.next_u32() & 0xFF for _ in range(16))

    def unit(self) -> float:
        """``rng.gen::<f64>()`` (rand 0.5 ``Standard``): 53 random bits scaled to [0, 1)."""
        return (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)

    def permutation(self, n: int) -> np.ndarray:
        """``Rng::shuffle`` applied to 0..n: ``i = n; while i >= 2: i -= 1; swap(i, gen_range(0, i + 1))``."""
        perm = np.arange(n, dtype=np.int64)
        i = n
        while i >= 2:
            i -= 1
            j = self.gen_range(0, i + 1)
            perm[i], perm[j] = perm[j], perm[i]
        return perm
