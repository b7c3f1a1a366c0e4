"""ctypes binding of the CPU oracle (oracle/libsbr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under sbr_rs_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from sbr_rs_amd._abi import SbrHparams, Status, storage_dim

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsbr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sbr_oracle.c")
    hdrs = [os.path.join(_HERE, "orc_numerics.h"), os.path.join(_HERE, "orc_ziggurat_tables.h"),
            os.path.join(_HERE, "..", "sbr_rs_amd", "csrc", "sbr_approx.h"), os.path.join(_HERE, "..", "include", "sbr_hip.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in [src] + hdrs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libsbr_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, u64p, u32p, fp = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        L.orc_model_create.argtypes = [C.POINTER(SbrHparams), C.POINTER(vp)]
        L.orc_model_destroy.argtypes = [vp]
        L.orc_model_destroy.restype = None
        L.orc_model_param_count.argtypes = [vp, C.c_int, u64p]
        L.orc_model_get_param.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.orc_model_set_param.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.orc_model_get_param_rows.argtypes = [vp, C.c_int, vp, C.c_uint64, vp]
        L.orc_model_get_epoch.argtypes = [vp]
        L.orc_model_get_epoch.restype = C.c_uint64
        L.orc_model_get_opt_steps.argtypes = [vp]
        L.orc_model_get_opt_steps.restype = C.c_uint64
        L.orc_chunk_lengths.argtypes = [C.c_uint64, C.c_uint64, u64p, C.c_int]
        L.orc_fit_begin.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp)]
        L.orc_fit_plan_destroy.argtypes = [vp]
        L.orc_fit_plan_destroy.restype = None
        L.orc_fit_epoch_prepare.argtypes = [vp, u64p]
        L.orc_fit_minibatch_rows.argtypes = [vp, C.c_int, C.c_uint64, u64p]
        L.orc_fit_last_offsets.argtypes = [vp, C.c_int, vp, C.c_uint64, u64p]
        L.orc_fit_step_local.argtypes = [vp, C.c_int, C.c_uint64]
        L.orc_fit_exchange_bytes.argtypes = [vp]
        L.orc_fit_exchange_bytes.restype = C.c_uint64
        L.orc_fit_export_local.argtypes = [vp, C.c_int, vp]
        L.orc_fit_step_apply.argtypes = [vp, vp]
        for name in ("orc_fit_chunk_bytes", "orc_fit_dense_bytes"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = C.c_uint64
        L.orc_fit_export_dense.argtypes = [vp, C.c_int, vp]
        L.orc_fit_scatter.argtypes = [vp, C.c_int, vp]
        L.orc_fit_owner_reduce.argtypes = [vp, vp, vp]
        L.orc_fit_apply_table.argtypes = [vp, vp, vp]
        L.orc_fit_owner_update.argtypes = [vp, C.c_int, vp]
        L.orc_fit_apply_dense_blocks.argtypes = [vp, vp]
        L.orc_model_table_slice_bytes.argtypes = [vp, C.c_int, u64p]
        L.orc_model_get_table_slice.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_model_set_table_slice.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_fit_step.argtypes = [vp, C.c_uint64]
        L.orc_fit_epoch_async.argtypes = [vp, C.c_uint64]
        L.orc_fit_end.argtypes = [vp, fp, u64p]
        L.orc_fit_end_lagged.argtypes = [vp, fp]
        L.orc_model_last_fit_lagged_loss.argtypes = [vp, fp]
        L.orc_fit_step_local_sample.argtypes = [vp, C.c_int, C.c_uint64, vp, C.c_uint32, vp, u32p, vp]
        L.orc_row_step.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, fp, fp]
        L.orc_row_reduce.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp, vp, fp, C.POINTER(C.c_int32)]
        L.orc_row_apply.argtypes = [vp, vp, C.c_float, C.c_int32, vp, vp, fp, fp]
        L.orc_dense_chain.argtypes = [vp, vp, C.c_uint64]
        L.orc_dense_chain.restype = C.c_float
        L.orc_model_apply_dense.argtypes = [vp, vp, C.c_uint64]
        L.orc_fit_threads.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_double, u64p, C.POINTER(C.c_double), fp]
        L.orc_model_padding_is_zero.argtypes = [vp]
        L.orc_model_set_reference_order.argtypes = [vp, C.c_int]
        L.orc_model_get_rng.argtypes = [vp, vp]
        L.orc_model_get_rng.restype = None
        L.orc_fit_debug_fetch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_uint64]
        L.orc_model_fit.argtypes = [vp, vp, vp, C.c_uint64, fp]
        L.orc_user_representation.argtypes = [vp, vp, C.c_uint64, vp]
        L.orc_predict.argtypes = [vp, vp, vp, C.c_uint64, vp]
        L.orc_mrr_score.argtypes = [vp, vp, vp, C.c_uint64, fp, vp, u64p]
        L.orc_rand_stream.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, vp, C.c_int]
        L.orc_rand_stream.restype = None
        L.orc_rand_uniform_u64.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int]
        L.orc_rand_uniform_u64.restype = C.c_uint64
        L.orc_tanh_pq.argtypes = [C.c_float, fp, fp]
        L.orc_tanh_pq.restype = None
        for name in ("orc_sigmoidf", "orc_tanhf", "orc_selftest_cell_h"):
            getattr(L, name).argtypes = [C.c_float]
            getattr(L, name).restype = C.c_float
        for name in ("orc_dot_tree", "orc_dot_chain"):
            getattr(L, name).argtypes = [vp, vp, C.c_int]
            getattr(L, name).restype = C.c_float
        L.orc_neg_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_neg_draw.restype = C.c_uint32
        L.orc_epoch_key.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_epoch_key.restype = C.c_uint64
        L.orc_adagrad.argtypes = [fp, fp, C.c_float, C.c_float, C.c_float]
        L.orc_adagrad.restype = None
        L.orc_adam.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_uint64]
        L.orc_adam.restype = None
        L.orc_xorshift_stream.argtypes = [vp, vp, C.c_int]
        L.orc_xorshift_stream.restype = None
        L.orc_fma_chain_gemm.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_fma_chain_gemm.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, status):
        self.status = Status(status)
        super().__init__(f"oracle status {self.status.name}")


def _check(st):
    if st != 0:
        raise OracleError(st)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


_DBG_DTYPE = {1: np.uint32, 7: np.uint32, 8: np.uint32, 9: np.uint32}


class OraclePlan:
    def __init__(self, model: "OracleModel", user_ptr, item_ids):
        self.model = model
        self._up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        self._it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().orc_fit_begin(model._h, _ptr(self._up), _ptr(self._it), len(self._up) - 1, C.byref(h)))
        self._h = h

    def epoch_prepare(self) -> int:
        n = C.c_uint64()
        _check(lib().orc_fit_epoch_prepare(self._h, C.byref(n)))
        return n.value

    def minibatch_rows(self, mb: int, device: int = 0) -> int:
        n = C.c_uint64()
        _check(lib().orc_fit_minibatch_rows(self._h, device, mb, C.byref(n)))
        return n.value

    def step(self, mb: int):
        _check(lib().orc_fit_step(self._h, mb))

    def last_offsets(self, device: int = 0) -> np.ndarray:
        """Packed layout of the device's last local step: off[t] = packed rows before step t, t = 0 .. Tm (sequences sorted by
        steps, descending; row of (step t, sequence b) = off[t] + b)."""
        T = int(self.model.hp.max_sequence_length)
        off = np.zeros(T + 1, dtype=np.uint64)
        tm = C.c_uint64()
        _check(lib().orc_fit_last_offsets(self._h, device, _ptr(off), off.size, C.byref(tm)))
        return off[: tm.value + 1].astype(np.int64)

    def exchange_bytes(self) -> int:
        return lib().orc_fit_exchange_bytes(self._h)

    def step_local(self, mb: int, device: int = 0) -> np.ndarray:
        """Compute device's local half-step and export its (single-device) exchange block."""
        _check(lib().orc_fit_step_local(self._h, device, mb))
        out = np.zeros(self.exchange_bytes(), dtype=np.uint8)
        _check(lib().orc_fit_export_local(self._h, device, _ptr(out)))
        return out

    def step_local_sample(self, mb: int, sel_b, device: int = 0):
        """Forward + negative sampling + loss + BPTT of ONLY the sequences `sel_b` (ascending indices into the full
        minibatch's packed order) with the position counters they have in the full minibatch.  Returns (rows, off): per
        compact packed row of the sample the packed row of the same (step, sequence) in the full minibatch, and the full
        minibatch's off table (rows before step t; T entries).  debug_fetch then reads the sample's arrays.  For parity at
        sizes the oracle cannot run whole."""
        sel = np.ascontiguousarray(sel_b, dtype=np.uint32)
        T = int(self.model.hp.max_sequence_length)
        rows = np.zeros(sel.size * (T - 1), dtype=np.uint32)
        off = np.zeros(T, dtype=np.uint64)
        n = C.c_uint32()
        _check(lib().orc_fit_step_local_sample(self._h, device, mb, _ptr(sel), sel.size, _ptr(rows), C.byref(n), _ptr(off)))
        return rows[: n.value].copy(), off

    def step_apply(self, all_blocks: np.ndarray):
        all_blocks = np.ascontiguousarray(all_blocks, dtype=np.uint8)
        _check(lib().orc_fit_step_apply(self._h, _ptr(all_blocks)))

    # ---- owner-reduce protocol (multi-device) ----
    def chunk_bytes(self) -> int:
        return lib().orc_fit_chunk_bytes(self._h)

    def dense_bytes(self) -> int:
        return lib().orc_fit_dense_bytes(self._h)

    def compute_local(self, mb: int, device: int):
        _check(lib().orc_fit_step_local(self._h, device, mb))

    def scatter(self, device: int, ndev: int) -> np.ndarray:
        out = np.zeros(ndev * self.chunk_bytes(), dtype=np.uint8)
        _check(lib().orc_fit_scatter(self._h, device, _ptr(out)))
        return out

    def export_dense(self, device: int) -> np.ndarray:
        out = np.zeros(self.dense_bytes(), dtype=np.uint8)
        _check(lib().orc_fit_export_dense(self._h, device, _ptr(out)))
        return out

    def owner_reduce(self, recv: np.ndarray) -> np.ndarray:
        recv = np.ascontiguousarray(recv, dtype=np.uint8)
        out = np.zeros(self.chunk_bytes(), dtype=np.uint8)
        _check(lib().orc_fit_owner_reduce(self._h, _ptr(recv), _ptr(out)))
        return out

    def apply_table(self, all_chunks: np.ndarray, dense_all: np.ndarray):
        all_chunks = np.ascontiguousarray(all_chunks, dtype=np.uint8)
        dense_all = np.ascontiguousarray(dense_all, dtype=np.uint8)
        _check(lib().orc_fit_apply_table(self._h, _ptr(all_chunks), _ptr(dense_all)))

    # ---- the owner-applied form (sbr_fit_step_owner_update): the owner updates its rows, parameter slices travel ----
    def owner_update(self, rank: int, recv: np.ndarray):
        recv = np.ascontiguousarray(recv, dtype=np.uint8)
        _check(lib().orc_fit_owner_update(self._h, rank, _ptr(recv)))

    def apply_dense_blocks(self, dense_all: np.ndarray):
        dense_all = np.ascontiguousarray(dense_all, dtype=np.uint8)
        _check(lib().orc_fit_apply_dense_blocks(self._h, _ptr(dense_all)))

    def end(self):
        loss, ex = C.c_float(), C.c_uint64()
        _check(lib().orc_fit_end(self._h, C.byref(loss), C.byref(ex)))
        return loss.value, ex.value

    def end_lagged(self) -> float:
        """The figure the reference's `fit` returns (stale loss-node values, sequence_model.rs:157 before :160)."""
        loss = C.c_float()
        _check(lib().orc_fit_end_lagged(self._h, C.byref(loss)))
        return loss.value

    def debug_fetch(self, which: int, rows: int, device: int = 0) -> np.ndarray:
        d = self.model.storage_dim
        which = int(which)
        if which in (0, 4, 5):
            out = np.zeros((rows, d), dtype=np.float32)
        elif which == 6:
            out = np.zeros(self.model.dense_count(), dtype=np.float32)
        elif which == 10:
            out = np.zeros((rows, {0: 4, 1: 3, 2: 0}[int(self.model.hp.model)] * d), dtype=np.float32)
        else:
            out = np.zeros(rows, dtype=_DBG_DTYPE.get(which, np.float32))
        _check(lib().orc_fit_debug_fetch(self._h, device, which, _ptr(out), out.nbytes))
        return out

    def close(self):
        if self._h:
            lib().orc_fit_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dense_chain(a, dz) -> np.float32:
    """One element of the dense gradient from its operand columns over all packed rows, contract order (a = None: bias row)."""
    dz = np.ascontiguousarray(dz, dtype=np.float32)
    if a is not None:
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == dz.size
    return np.float32(lib().orc_dense_chain(_ptr(a) if a is not None else None, _ptr(dz), dz.size))


def row_reduce(vecs, scale, has_bias):
    """One DEVICE's gradient of one item-table row from its ordered entry list (the contract's chunked in-order sum):
    (g [d], bias gradient, has_bias)."""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    hb = np.ascontiguousarray(has_bias, dtype=np.uint8)
    g = np.zeros(vecs.shape[1], dtype=np.float32)
    gb, has = C.c_float(), C.c_int32()
    _check(lib().orc_row_reduce(vecs.shape[1], scale.size, _ptr(vecs), _ptr(scale), _ptr(hb), _ptr(g), C.byref(gb), C.byref(has)))
    return g, np.float32(gb.value), bool(has.value)


class OracleModel:
    def __init__(self, hp: SbrHparams):
        self.hp = hp
        self.dim = int(hp.embedding_dim)
        h = C.c_void_p()
        _check(lib().orc_model_create(C.byref(hp), C.byref(h)))
        self._h = h
        self.storage_dim = storage_dim(self.dim)

    def dense_count(self) -> int:
        d = self.storage_dim
        ng = {0: 4, 1: 3, 2: 0}[int(self.hp.model)]
        return (2 * d + 1) * ng * d if ng else d

    def param_count(self, which: int) -> int:
        n = C.c_uint64()
        _check(lib().orc_model_param_count(self._h, int(which), C.byref(n)))
        return n.value

    def get_param(self, which: int) -> np.ndarray:
        out = np.zeros(self.param_count(which), dtype=np.float32)
        _check(lib().orc_model_get_param(self._h, int(which), _ptr(out), out.size))
        return out

    def get_param_rows(self, which: int, rows) -> np.ndarray:
        """Selected rows of an item-table block ([n, embedding_dim]; biases: [n]) without copying the whole table."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        table = self.param_count(which) != int(self.hp.num_items)
        out = np.zeros((rows.size, self.dim) if table else rows.size, dtype=np.float32)
        _check(lib().orc_model_get_param_rows(self._h, int(which), _ptr(rows), rows.size, _ptr(out)))
        return out

    def set_param(self, which: int, values: np.ndarray):
        values = np.ascontiguousarray(values, dtype=np.float32).ravel()
        _check(lib().orc_model_set_param(self._h, int(which), _ptr(values), values.size))

    def table_slice(self, which: int, rank: int) -> np.ndarray:
        """Owner slice `rank` of an item-table block in stored layout (zero-padded to ceil(I / num_devices) rows), as bytes."""
        n = C.c_uint64()
        _check(lib().orc_model_table_slice_bytes(self._h, int(which), C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(lib().orc_model_get_table_slice(self._h, int(which), rank, _ptr(out)))
        return out

    def set_table_slice(self, which: int, rank: int, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        _check(lib().orc_model_set_table_slice(self._h, int(which), rank, _ptr(data)))

    def global_epoch(self) -> int:
        return lib().orc_model_get_epoch(self._h)

    def padding_is_zero(self) -> bool:
        """Every stored element beyond embedding_dim (parameters and optimiser state) is still zero."""
        return bool(lib().orc_model_padding_is_zero(self._h))

    def optimizer_steps(self) -> int:
        return lib().orc_model_get_opt_steps(self._h)

    def get_rng(self) -> bytes:
        """The model RNG's state as the 16 bytes that re-create it through XorShiftRng::from_seed."""
        out = (C.c_uint8 * 16)()
        lib().orc_model_get_rng(self._h, out)
        return bytes(out)

    def set_reference_order(self, on: bool = True) -> None:
        """Checker-only mode (batch_sequences = 1): negatives from the partition's sequential xorshift stream
        (sequence_model.rs:58-65, 137) and, with num_devices > 1, one optimiser application per device in device order
        (wyrm's SynchronizedOptimizer as recalled) — the reference's order of work in the two places where the
        engine's contract substitutes its own (counter-keyed draws; one update from the summed gradients)."""
        _check(lib().orc_model_set_reference_order(self._h, 1 if on else 0))

    def fit(self, user_ptr, item_ids) -> float:
        up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        loss = C.c_float()
        _check(lib().orc_model_fit(self._h, _ptr(up), _ptr(it), len(up) - 1, C.byref(loss)))
        return loss.value

    def row_step(self, vecs, scale, has_bias, w, acc, b, bacc):
        """One item-table row's optimiser step from an explicit, ordered entry list (contract order of the per-row reduction,
        then Adagrad with this model's hyper-parameters): returns (w, acc, b, bacc) after the step."""
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        scale = np.ascontiguousarray(scale, dtype=np.float32)
        hb = np.ascontiguousarray(has_bias, dtype=np.uint8)
        w = np.array(w, dtype=np.float32, copy=True)
        acc = np.array(acc, dtype=np.float32, copy=True)
        bb, ba = C.c_float(float(b)), C.c_float(float(bacc))
        assert vecs.shape == (scale.size, self.storage_dim) and w.size == acc.size == self.storage_dim
        _check(lib().orc_row_step(self._h, scale.size, _ptr(vecs), _ptr(scale), _ptr(hb), _ptr(w), _ptr(acc), C.byref(bb), C.byref(ba)))
        return w, acc, np.float32(bb.value), np.float32(ba.value)

    def row_apply(self, g, gb, has_bias, w, acc, b, bacc):
        """The optimiser update (Adagrad with this model's hyper-parameters) of one item-table row from an explicit gradient:
        returns (w, acc, b, bacc) after the step."""
        g = np.ascontiguousarray(g, dtype=np.float32)
        w = np.array(w, dtype=np.float32, copy=True)
        acc = np.array(acc, dtype=np.float32, copy=True)
        bb, ba = C.c_float(float(b)), C.c_float(float(bacc))
        _check(lib().orc_row_apply(self._h, _ptr(g), C.c_float(float(gb)), 1 if has_bias else 0, _ptr(w), _ptr(acc), C.byref(bb), C.byref(ba)))
        return w, acc, np.float32(bb.value), np.float32(ba.value)

    def apply_dense(self, dense):
        """The dense half of one optimiser step from an explicit dense-gradient block."""
        dense = np.ascontiguousarray(dense, dtype=np.float32)
        _check(lib().orc_model_apply_dense(self._h, _ptr(dense), dense.size))

    def fit_threads(self, user_ptr, item_ids, workers: int, synchronous: bool, max_seconds: float = 0.0):
        """The reference's parallel shape (sequence_model.rs:90-102) for the timed CPU baseline: `workers` threads on THIS
        model's one shared parameter set, one partition each; Hogwild (no locks) or barrier-synchronised optimiser steps.
        Not deterministic.  Returns (interactions processed, wall seconds, loss)."""
        up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        rows, secs, loss = C.c_uint64(), C.c_double(), C.c_float()
        _check(lib().orc_fit_threads(self._h, _ptr(up), _ptr(it), len(up) - 1, int(workers), 1 if synchronous else 0,
                                     float(max_seconds), C.byref(rows), C.byref(secs), C.byref(loss)))
        return rows.value, secs.value, loss.value

    def last_fit_lagged_loss(self) -> float:
        """The figure the reference's `fit` would have returned for the last `fit` call."""
        v = C.c_float()
        _check(lib().orc_model_last_fit_lagged_loss(self._h, C.byref(v)))
        return v.value

    def fit_begin(self, user_ptr, item_ids) -> OraclePlan:
        return OraclePlan(self, user_ptr, item_ids)

    def user_representation(self, item_ids) -> np.ndarray:
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        out = np.zeros(self.dim, dtype=np.float32)
        _check(lib().orc_user_representation(self._h, _ptr(it), it.size, _ptr(out)))
        return out

    def predict(self, user, item_ids) -> np.ndarray:
        user = np.ascontiguousarray(user, dtype=np.float32)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        out = np.zeros(it.size, dtype=np.float32)
        _check(lib().orc_predict(self._h, _ptr(user), _ptr(it), it.size, _ptr(out)))
        return out

    def mrr_score(self, user_ptr, item_ids):
        up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        ranks = np.zeros(len(up) - 1, dtype=np.uint32)
        mrr, n = C.c_float(), C.c_uint64()
        _check(lib().orc_mrr_score(self._h, _ptr(up), _ptr(it), len(up) - 1, C.byref(mrr), _ptr(ranks), C.byref(n)))
        return mrr.value, ranks[: n.value].copy()

    def close(self):
        if self._h:
            lib().orc_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fma_chain_gemm(a: np.ndarray, b: np.ndarray, c0=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    M, K = a.shape
    N = b.shape[1]
    out = np.zeros((M, N), dtype=np.float32)
    c0p = None
    if c0 is not None:
        c0 = np.ascontiguousarray(c0, dtype=np.float32)
        c0p = _ptr(c0)
    lib().orc_fma_chain_gemm(_ptr(a), _ptr(b), c0p, M, K, N, _ptr(out))
    return out
