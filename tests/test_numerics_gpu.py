"""GPU: the numerics contract (sbr_rs_amd/csrc/sbr_numerics.h) evaluates to the same bits on the
gfx950 device as in the CPU oracle.  These are the premises every parity claim rests on:
IEEE +,*,/,sqrt,fma on both sides; the wave butterfly equals the oracle's tree; an f32 MFMA
accumulation equals a k-ascending fmaf chain."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def L():
    from sbr_rs_amd import _lib

    return _lib.load()


def test_math_bitwise(L, oracle_lib):
    rs = np.random.RandomState(1)
    x = np.concatenate([
        rs.uniform(-100, 100, 200000), rs.uniform(-1, 1, 200000), rs.normal(0, 1e-3, 50000),
        np.array([0.0, -0.0, 0.625, -0.625, 0.6249999, 88.0, -87.0, 1e-30, -1e-30, 1e30, -1e30, 87.9999, -86.9999]),
    ]).astype(np.float32)
    n = x.size
    e, s, t = (np.zeros(n, np.float32) for _ in range(3))
    assert L.sbr_selftest_math(_p(x), n, _p(e), _p(s), _p(t)) == 0
    ce = np.array([oracle_lib.orc_selftest_cell_h(float(v)) for v in x], dtype=np.float32)
    cs = np.array([oracle_lib.orc_sigmoidf(float(v)) for v in x], dtype=np.float32)
    ct = np.array([oracle_lib.orc_tanhf(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(e.view(np.uint32), ce.view(np.uint32))
    assert np.array_equal(s.view(np.uint32), cs.view(np.uint32))
    assert np.array_equal(t.view(np.uint32), ct.view(np.uint32))
    # a NaN pre-activation stays NaN through the clamp (diverged weights must not yield finite-looking states)
    bad = np.array([np.nan, 1.0, -np.nan, np.inf], dtype=np.float32)
    e2, s2, t2 = (np.zeros(4, np.float32) for _ in range(3))
    assert L.sbr_selftest_math(_p(bad), 4, _p(e2), _p(s2), _p(t2)) == 0
    assert np.isnan(t2[0]) and np.isnan(s2[0]) and np.isnan(e2[0]) and np.isnan(t2[2]) and not np.isnan(t2[1])
    assert t2[3] == np.float32(oracle_lib.orc_tanhf(float("inf"))) and np.isnan(np.float32(oracle_lib.orc_tanhf(float("nan"))))
    # and the device values themselves against float64 libm (the oracle shares the polynomial, so this
    # is the check that the polynomial is a tanh): 3e-7 for tanh / sigmoid
    x64 = x.astype(np.float64)
    assert np.max(np.abs(t.astype(np.float64) - np.tanh(x64))) < 3e-7
    assert np.max(np.abs(s.astype(np.float64) - 1 / (1 + np.exp(-x64)))) < 3e-7


@pytest.mark.parametrize("d", [16, 32, 64, 128, 256])
def test_dot_tree_bitwise(L, oracle_lib, d):
    rs = np.random.RandomState(d)
    n = 1000 + d  # not a multiple of the rows-per-wave
    x = rs.normal(0, 1, (n, d)).astype(np.float32)
    y = rs.normal(0, 1, (n, d)).astype(np.float32)
    out = np.zeros(n, np.float32)
    assert L.sbr_selftest_dot_tree(_p(x), _p(y), d, n, _p(out)) == 0
    ref = np.array([oracle_lib.orc_dot_tree(_p(x[i]), _p(y[i]), d) for i in range(n)], dtype=np.float32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("k", [4, 64, 512, 2048])
def test_mfma_is_fma_chain(L, k):
    rs = np.random.RandomState(k)
    a = rs.normal(0, 1, (16, k)).astype(np.float32)
    b = rs.normal(0, 1, (k, 16)).astype(np.float32)  # asymmetric on purpose
    c0 = rs.normal(0, 1, (16, 16)).astype(np.float32)
    a32 = rs.normal(0, 1, (32, k)).astype(np.float32)
    b32 = rs.normal(0, 1, (k, 32)).astype(np.float32)
    out = np.zeros((16, 16), np.float32)
    out32 = np.zeros((32, 32), np.float32)
    assert L.sbr_selftest_mfma(_p(a), _p(b), _p(c0), k, _p(out), _p(a32), _p(b32), _p(out32)) == 0
    ref = oracle.fma_chain_gemm(a, b, c0)
    ref32 = oracle.fma_chain_gemm(a32, b32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(out32.view(np.uint32), ref32.view(np.uint32))


@pytest.mark.parametrize("n,row_bits,dist", [
    (1, 11, "uniform"), (63, 11, "uniform"), (381, 11, "uniform"), (4096, 11, "uniform"), (4097, 6, "uniform"),
    (100_000, 11, "zipf"), (300_001, 20, "uniform"), (1_500_000, 20, "zipf"), (700_000, 24, "uniform"),
    (50_000, 32, "uniform"), (20_000, 3, "uniform"), (123_457, 17, "one_row"),
])
@pytest.mark.parametrize("small,items", [("1", None), ("0", None), ("0", "64")])
def test_key_ordering_is_a_stable_sort_by_row(L, monkeypatch, n, row_bits, dist, small, items):
    """sbr_sort.hip against numpy's stable argsort: keys (row << 32 | e) in (row, e) order — one, two and three radix passes,
    ragged last tiles, hot rows, a single row — and the ascending list of segment heads with its sentinel.  Inputs of up to
    4 096 keys take the single-launch LDS-resident form unless SBR_SORT_SMALL=0 sends them through the tiled passes; the
    tiled passes cut up to 2^18 keys into 512-key tiles and more into 4 096-key tiles (SBR_SORT_ITEMS=64: always)."""
    monkeypatch.setenv("SBR_SORT_SMALL", small)
    if items:
        monkeypatch.setenv("SBR_SORT_ITEMS", items)
    rs = np.random.RandomState(n % 9973 + row_bits)
    hi = (1 << row_bits) if row_bits < 32 else (1 << 32) - 1
    if dist == "uniform":
        rows = rs.randint(0, hi, size=n, dtype=np.int64)
    elif dist == "zipf":
        rows = np.minimum(rs.zipf(1.3, size=n) - 1, hi - 1)
    else:
        rows = np.full(n, hi - 1, dtype=np.int64)
    rows = rows.astype(np.uint32)
    keys = np.zeros(n, np.uint64)
    heads = np.zeros(n + 1, np.uint32)
    nheads = np.zeros(1, np.uint32)
    assert L.sbr_selftest_sort(_p(rows), n, row_bits, _p(keys), _p(heads), _p(nheads)) == 0
    order = np.argsort(rows, kind="stable")
    want = (rows[order].astype(np.uint64) << np.uint64(32)) | order.astype(np.uint64)
    assert np.array_equal(keys, want)
    srt = rows[order]
    want_heads = np.flatnonzero(np.concatenate([[True], srt[1:] != srt[:-1]])).astype(np.uint32)
    assert int(nheads[0]) == want_heads.size
    assert np.array_equal(heads[: want_heads.size], want_heads)
    assert int(heads[want_heads.size]) == n
