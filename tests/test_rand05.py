"""CPU: the rand 0.5 generators as recalled — the three statements of them (the oracle's
C, the Python host mirror, the C++ host layer is covered by tests/test_cpp_facade.py through the engine)
agree with each other and with independent restatements written here; the ziggurat tables regenerate to
the crate's published leading entries; the normal sampler is a standard normal."""
import ctypes as C
import os
import re

import numpy as np

from sbr_rs_amd.rng import XorShiftRng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


def _seed(k):
    return bytes((k * 37 + i * 11 + 1) & 0xFF for i in range(16))


def _stream(lib, seed, what, n, a=0, b=0):
    out = np.zeros(n, dtype=np.float64)
    s = np.frombuffer(seed, dtype=np.uint8).copy()
    lib.orc_rand_stream(s.ctypes.data_as(C.c_void_p), what, a, b, out.ctypes.data_as(C.c_void_p), n)
    return out


def test_gen_range_matches_independent_restatement(oracle_lib):
    """UniformInt::sample_single: zone = range << leading_zeros(range); v = next_u64; (hi, lo) = v * range;
    accept lo <= zone.  For range = 5 the zone is 5 << 61 = 0.625 * 2^64: 3 draws in 8 are rejected."""
    for n in (1, 2, 5, 1321, 1 << 33, (1 << 63) + 5):
        seed = _seed(n % 251)
        r = XorShiftRng.from_seed(seed)
        want = []
        ref = XorShiftRng.from_seed(seed)
        zone = (n << (64 - n.bit_length())) & M64
        for _ in range(200):
            while True:
                m = ref.next_u64() * n
                if (m & M64) <= zone:
                    want.append(m >> 64)
                    break
        got_py = [r.gen_range(0, n) for _ in range(200)]
        got_c = _stream(oracle_lib, seed, 0, 200, n)
        assert got_py == want
        if n < (1 << 53):
            assert [int(v) for v in got_c] == want
        assert all(0 <= v < n for v in want)
    # rejection really happens for range 5
    ref = XorShiftRng.from_seed(_seed(3))
    rej = sum(1 for _ in range(4000) if ((ref.next_u64() * 5) & M64) > (5 << 61))
    assert 1300 < rej < 1700


def test_uniform_new_matches_and_split_keys(oracle_lib):
    """Uniform::new(0, u64::MAX): range = 2^64 - 1, one value rejected in 2^64 — the key is v - 1 for any
    draw v >= 1 (user_based_split, data.rs:77-78)."""
    seed = bytes([42] * 16)
    r = XorShiftRng.from_seed(seed)
    ref = XorShiftRng.from_seed(seed)
    for skip in range(2):
        v = ref.next_u64()
        key = r.uniform(0, M64)
        assert key == v - 1
        s = np.frombuffer(seed, dtype=np.uint8).copy()
        assert oracle_lib.orc_rand_uniform_u64(s.ctypes.data_as(C.c_void_p), 0, M64, skip) == key
    for lo, hi in ((0, 1683), (3, 7), (10, 1 << 40)):
        a = XorShiftRng.from_seed(_seed(9))
        got = [a.uniform(lo, hi) for _ in range(300)]
        assert all(lo <= g < hi for g in got)
        assert [int(v) for v in _stream(oracle_lib, _seed(9), 1, 300, lo, hi)] == got


def test_gen_seed_is_sixteen_truncated_words(oracle_lib):
    seed = _seed(5)
    ref = XorShiftRng.from_seed(seed)
    want = bytes(ref.next_u32() & 0xFF for _ in range(16))
    assert XorShiftRng.from_seed(seed).gen_seed() == want
    got = _stream(oracle_lib, seed, 3, 1)  # first byte of the first generated seed
    assert int(got[0]) == want[0]


def test_shuffle_is_fisher_yates_from_the_end():
    r = XorShiftRng.from_seed(_seed(1))
    ref = XorShiftRng.from_seed(_seed(1))
    n = 50
    want = list(range(n))
    i = n
    while i >= 2:
        i -= 1
        j = ref.gen_range(0, i + 1)
        want[i], want[j] = want[j], want[i]
    assert list(r.permutation(n)) == want and sorted(want) == list(range(n))


def test_ziggurat_tables_regenerate_and_headers_agree():
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_zig", os.path.join(ROOT, "tools", "gen_ziggurat_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    x, f = mod.tables()
    # the crate's published leading entries (ziggurat_tables.rs), 18 decimals
    assert ["%.18f" % v for v in x[:4]] == ["3.910757959537090045", "3.654152885361008796", "3.449278298560964462",
                                            "3.320244733839166074"]
    assert len(x) == 257 and x[256] == 0.0 and f[256] == 1.0 and all(x[i] > x[i + 1] for i in range(256))
    # every layer has the same area V = x[i] * (f[i+1] - f[i]) (i >= 1), the defining property
    areas = [x[i] * (f[i + 1] - f[i]) for i in range(1, 256)]
    assert max(abs(a - mod.V) for a in areas[:-1]) < 1e-13
    tabs = []
    for path in ("sbr_rs_amd/csrc/sbr_ziggurat_tables.h", "oracle/orc_ziggurat_tables.h"):
        text = open(os.path.join(ROOT, path)).read()
        vals = [float.fromhex(h) for h in re.findall(r"0x[0-9a-f.]+p[+-]\d+", text)]
        assert len(vals) == 1 + 2 * 257
        tabs.append(vals)
    assert tabs[0] == tabs[1] == [mod.R] + x + f


def test_standard_normal_moments_and_tail(oracle_lib):
    z = _stream(oracle_lib, _seed(77), 2, 400000)
    assert abs(z.mean()) < 6e-3 and abs(z.std() - 1.0) < 5e-3
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3.0) < 0.06
    assert 0.0020 < np.mean(np.abs(z) > 3.0) < 0.0034  # 0.0027
    assert np.abs(z).max() > 3.66  # the tail beyond R = 3.654 is reached


def test_empty_range_is_rejected_like_the_crate():
    """rand 0.5 asserts low < high in gen_range / Uniform::new; the mirrors raise instead of shifting by 64."""
    import pytest

    r = XorShiftRng.from_seed(_seed(3))
    before = r.state_seed()
    for bad in ((0, 0), (5, 5), (7, 3)):
        with pytest.raises(ValueError):
            r.gen_range(*bad)
        with pytest.raises(ValueError):
            r.uniform(*bad)
    with pytest.raises(ValueError):
        r.below(0)
    assert r.state_seed() == before  # nothing was drawn
