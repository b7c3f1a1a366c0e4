"""The C++ host layer (include/sbr.hpp) over the C-ABI: the reference crate's own tests for the
sequence-model path, written in C++ (tests/cpp/facade_tests.cpp), driven from here.

CPU part: the data-side tests (data.rs:587-660), the RNG / SipHash streams and the MovieLens
split must agree with the Python host layer value for value, and `build()` without a GPU must
raise (no CPU fallback).  GPU part: lstm.rs:451-530 / ewma.rs:455-507 run through the façade and
are compared bit for bit with the CPU oracle (loss to 1e-6: its summation order is free)."""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

from helpers import LOSS_HINGE, LOSS_WARP, hparams, load_movielens, movielens_protocol
from sbr_rs_amd import build as hip_build
from sbr_rs_amd._abi import ModelKind
from sbr_rs_amd.data import _siphash24_u64
from sbr_rs_amd.rng import XorShiftRng


@pytest.fixture(scope="module")
def facade():
    hip_build.build(verbose=False)
    return hip_build.build_facade_tests(verbose=False)


def run(binary, *args, check=True):
    p = subprocess.run([binary, *args], capture_output=True, text=True, timeout=900)
    if check:
        assert p.returncode == 0, (args, p.returncode, p.stdout, p.stderr)
    out = {}
    for line in p.stdout.splitlines():
        for tok in line.split():
            if "=" in tok:
                k, v = tok.split("=", 1)
                out[k] = v
    return p.returncode, out


def fnv(a: np.ndarray) -> int:
    h = 1469598103934665603
    for b in a.tobytes():
        h = ((h ^ b) * 1099511628211) & ((1 << 64) - 1)
    return h


@pytest.fixture(scope="module")
def movielens_csv(tmp_path_factory):
    """The fixture re-serialised in the reference's CSV layout (datasets.rs:57-60)."""
    data = load_movielens()
    users, items, ts = data.arrays()
    path = tmp_path_factory.mktemp("ml") / "data.csv"
    with open(path, "w") as f:
        f.write("user_id,item_id,rating,timestamp\n")
        for u, i, t in zip(users, items, ts):
            f.write(f"{int(u)},{int(i)},1,{int(t)}\n")
    return str(path)


# ---- no GPU needed ----------------------------------------------------------------------------
def test_reference_data_tests_in_cpp(facade):
    run(facade, "to_compressed")        # data.rs:587-627
    run(facade, "test_chunk_iterator")  # data.rs:629-660
    run(facade, "triplet_minibatches")  # data.rs:435-575


def test_streams_match_python_host_layer(facade):
    _, o = run(facade, "streams")
    r = XorShiftRng.from_seed(bytes([42] * 16))
    assert o["u32"] == ",".join(str(r.next_u32()) for _ in range(3))
    assert o["u64"] == str(r.next_u64())
    assert o["below"] == ",".join(str(r.below(n)) for n in (1683, 1000000, (1 << 64) - 1))
    assert float(o["unit"]) == r.unit()
    assert o["shuffle"] == ",".join(str(int(v)) for v in np.arange(10)[r.permutation(10)])
    assert o["state"] == r.state_seed().hex()
    sip = [int(_siphash24_u64(0x0706050403020100, 0x0F0E0D0C0B0A0908, np.array([0], dtype=np.uint64))[0]),
           int(_siphash24_u64(1, 2, np.array([943], dtype=np.uint64))[0])]
    assert o["siphash"] == ",".join(map(str, sip))
    assert o["zero_seed_u32"] == str(XorShiftRng.from_seed(bytes(16)).next_u32())


def test_movielens_split_matches_python_host_layer(facade, movielens_csv):
    """CSV reader + user_based_split + to_compressed: same CSR, same RNG state afterwards."""
    _, o = run(facade, "split", movielens_csv)
    data, train, test, rng = movielens_protocol()
    assert (int(o["num_users"]), int(o["num_items"]), int(o["len"])) == (data.num_users(), data.num_items(), data.len())
    assert int(o["train_nnz"]) == len(train.item_ids) and int(o["test_nnz"]) == len(test.item_ids)
    assert int(o["train_ptr_hash"]) == fnv(train.user_pointers) and int(o["train_items_hash"]) == fnv(train.item_ids)
    assert int(o["test_ptr_hash"]) == fnv(test.user_pointers) and int(o["test_items_hash"]) == fnv(test.item_ids)
    assert o["rng_state"] == rng.state_seed().hex()


def test_build_without_device_raises(facade):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _, o = run(facade, "no_device")
    assert o == {"engine_error": "5"}  # SBR_ERR_NO_DEVICE: never a CPU fallback


# ---- on the GPU -------------------------------------------------------------------------------
@pytest.mark.gpu
def test_empty_interactions_and_defaults(facade):
    run(facade, "empty_interactions")    # lstm.rs:520-530
    run(facade, "defaults_and_predict")
    run(facade, "partitioned_equals_replicated")


@pytest.mark.gpu
def test_save_load_roundtrip_in_cpp(facade, tmp_path):
    """≙ the serde derives: `model.save(path)` / `Implicit*Model::load(path)` of the C++ host layer — parameters,
    optimiser state (Adam moments included), counters and RNG; one replica and three replicas over a partitioned table."""
    assert run(facade, "save_load_roundtrip", str(tmp_path))[1].get("save_load") == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("case,kind,loss,threads,T,B", [
    ("mrr_test_single_thread", ModelKind.LSTM_NORMAL, LOSS_HINGE, 1, 128, 8),  # lstm.rs:451-473
    ("mrr_test_two_threads", ModelKind.LSTM_NORMAL, LOSS_HINGE, 2, 128, 8),    # lstm.rs:475-497
    ("mrr_test_warp", ModelKind.LSTM_NORMAL, LOSS_WARP, 1, 128, 8),            # lstm.rs:499-520
    ("mrr_test_ewma", ModelKind.EWMA, LOSS_HINGE, 1, 128, 8),                  # ewma.rs:455-487
    ("crate_doctest", ModelKind.LSTM_NORMAL, LOSS_WARP, 1, 32, 32),            # lib.rs:22-58 (builder-default minibatch)
])
def test_reference_mrr_tests_in_cpp_match_oracle(facade, movielens_csv, oracle_lib, case, kind, loss, threads, T, B):
    from oracle.oracle import OracleModel

    _, o = run(facade, case, movielens_csv)
    data, train, test, rng = movielens_protocol()
    hp = hparams(data.num_items(), T, 32, int(kind), loss, epochs=10, B=B, seed=rng.state_seed(), ndev=threads)
    orc = OracleModel(hp)
    loss_o = orc.fit(train.user_pointers, train.item_ids)
    mrr_o, ranks_o = orc.mrr_score(test.user_pointers, test.item_ids)
    assert int(o["ranks"]) == len(ranks_o) and int(o["ranks_hash"]) == fnv(np.asarray(ranks_o, dtype=np.uint32))
    assert o["test_mrr_bits"] == f"{np.float32(mrr_o).view(np.uint32):08x}"
    assert float(o["loss"]) == pytest.approx(loss_o, rel=1e-6)
