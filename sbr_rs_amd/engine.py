"""Thin object layer over the C-ABI (include/sbr_hip.h).  Mirrors oracle/oracle.py's shape so
parity tests read symmetrically, but talks only to libsbr_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._abi import NUM_KERNEL_FAMILIES, KernelFamily, SbrHparams, Status, storage_dim
from .errors import EngineError, FittingError, PredictionError


def _check(st: int):
    if st == Status.OK:
        return
    msg = _lib.load().sbr_status_string(st).decode()
    if st == Status.NO_INTERACTIONS:
        raise FittingError.NoInteractions(msg)
    if st == Status.INVALID_PREDICTION:
        raise PredictionError.InvalidPredictionValue(msg)
    raise EngineError(Status(st), msg)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_info():
    L = _lib.load()
    name = C.create_string_buffer(64)
    cus, hbm = C.c_uint32(), C.c_uint64()
    _check(L.sbr_device_info(name, 64, C.byref(cus), C.byref(hbm)))
    return name.value.decode(), cus.value, hbm.value


def set_device(ordinal: int):
    _check(_lib.load().sbr_set_device(int(ordinal)))


def device_count() -> int:
    n = C.c_int32()
    _check(_lib.load().sbr_device_count(C.byref(n)))
    return n.value


GROUP_PARTITION_ITEM_TABLE = 1


def group_create(hp: SbrHparams, n: int, partition_item_table: bool = False):
    """The n replicas of a single-process group (sbr_group_create): replica r on HIP device r mod device
    count.  ``partition_item_table``: the item table exists once, row range r on replica r's device,
    mapped into every replica (BASELINE configs[4]); results equal the replicated group bit for bit."""
    import copy

    L = _lib.load()
    handles = (C.c_void_p * n)()
    _check(L.sbr_group_create(C.byref(hp), n, GROUP_PARTITION_ITEM_TABLE if partition_item_table else 0, handles))
    out = []
    for r in range(n):
        h = copy.copy(hp)
        h.num_devices, h.device_rank = n, r
        out.append(Model._from_handle(h, C.c_void_p(handles[r])))
    return out


def group_fit(models, user_ptr, item_ids) -> float:
    """Single-process multi-device fit (sbr_group_fit): models[r] built with num_devices = len(models),
    device_rank = r and the same seed.  ≙ fit with num_threads(n) in one process."""
    L = _lib.load()
    up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
    it = np.ascontiguousarray(item_ids, dtype=np.uint32)
    handles = (C.c_void_p * len(models))(*[m._h for m in models])
    loss = C.c_float()
    _check(L.sbr_group_fit(handles, len(models), _ptr(up), _ptr(it), len(up) - 1, C.byref(loss)))
    return loss.value


class GroupPlan:
    """sbr_group_fit taken apart (sbr_group_fit_begin .. sbr_group_fit_end): the single-process group's steps one at a time —
    the bench's ``--driver group`` and the parity tests of the multi-device step.  ``host_threads``: None = the library's
    default (one host thread per device from four devices on), True / False = force."""

    def __init__(self, models, user_ptr, item_ids, host_threads=None):
        self.models = list(models)
        self._L = _lib.load()
        self._up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        self._it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        handles = (C.c_void_p * len(self.models))(*[m._h for m in self.models])
        h = C.c_void_p()
        _check(self._L.sbr_group_fit_begin(handles, len(self.models), _ptr(self._up), _ptr(self._it), len(self._up) - 1, C.byref(h)))
        self._h = h
        if host_threads is not None:
            _check(self._L.sbr_group_plan_set_host_threads(self._h, 1 if host_threads else 0))
        self._members = {}

    def epoch_prepare(self, prefetch_next: bool = False) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_group_epoch_prepare(self._h, C.byref(n), 1 if prefetch_next else 0))
        return n.value

    def step(self, mb: int):
        _check(self._L.sbr_group_step(self._h, mb))

    def step_local(self, mb: int):
        _check(self._L.sbr_group_step_local(self._h, mb))

    def member(self, r: int) -> "FitPlan":
        """Replica r's plan, borrowed from the group (debug_fetch, minibatch_rows, counters)."""
        if r not in self._members:
            h = C.c_void_p()
            _check(self._L.sbr_group_member_plan(self._h, r, C.byref(h)))
            self._members[r] = FitPlan._borrowed(self.models[r], h)
        return self._members[r]

    def synchronize(self):
        _check(self._L.sbr_group_synchronize(self._h))

    def set_exchange(self, gradient_all_gather: bool):
        """Synchronous, replicated: False (default) = owner-applied update, parameter slices gathered in place; True = rounds 1-5's
        gradient all-gather + whole-table update on every replica (sbr_group_plan_set_exchange).  Same bits."""
        _check(self._L.sbr_group_plan_set_exchange(self._h, 1 if gradient_all_gather else 0))

    def gather_optimizer_state(self):
        """Every replica's item-table optimiser state made complete from the owners' slices (fit end does it)."""
        _check(self._L.sbr_group_gather_optimizer_state(self._h))

    def stats(self):
        """(host ms spent queueing steps, steps, host threads in use)."""
        ms, n, th = C.c_double(), C.c_uint64(), C.c_int32()
        _check(self._L.sbr_group_plan_stats(self._h, C.byref(ms), C.byref(n), C.byref(th)))
        return ms.value, n.value, th.value

    def end(self) -> float:
        loss = C.c_float()
        h, self._h = self._h, None
        self._members = {}
        _check(self._L.sbr_group_fit_end(h, C.byref(loss)))
        return loss.value

    def close(self):
        if getattr(self, "_h", None):
            self._members = {}
            self._L.sbr_group_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL inside the library (sbr_comm_*): for hosts with one process per GPU and no collective library of their own.
    ``Comm.unique_id()`` on rank 0, the 128 bytes to every rank by any channel, ``Comm(id, world, rank)`` on every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(_lib.load().sbr_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes, world: int, rank: int):
        if len(uid) != 128:
            raise ValueError("the id is 128 bytes")
        self._L = _lib.load()
        h = C.c_void_p()
        _check(self._L.sbr_comm_create((C.c_uint8 * 128)(*uid), world, rank, C.byref(h)))
        self._h, self.world, self.rank = h, world, rank

    def fit(self, model: "Model", user_ptr, item_ids) -> float:
        """This rank's whole fit through the library's own transport (sbr_model_fit_comm)."""
        up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        loss = C.c_float()
        _check(self._L.sbr_model_fit_comm(model._h, self._h, _ptr(up), _ptr(it), len(up) - 1, C.byref(loss)))
        return loss.value

    def step_exchange(self, plan: "FitPlan", mb: int):
        _check(self._L.sbr_fit_step_exchange(plan._h, mb, self._h))

    def gather_optimizer_state(self, model: "Model"):
        _check(self._L.sbr_comm_gather_optimizer_state(model._h, self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.sbr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DBG_U32 = {1, 7, 8, 9}


class FitPlan:
    def __init__(self, model: "Model", user_ptr, item_ids):
        self.model = model
        self._L = _lib.load()
        self._up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        self._it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        h = C.c_void_p()
        _check(self._L.sbr_fit_begin(model._h, _ptr(self._up), _ptr(self._it), len(self._up) - 1, C.byref(h)))
        self._h = h

    @classmethod
    def _borrowed(cls, model: "Model", handle) -> "FitPlan":
        p = cls.__new__(cls)
        p.model, p._L, p._h, p._owned = model, _lib.load(), handle, False
        return p

    def epoch_prepare(self) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_fit_epoch_prepare(self._h, C.byref(n)))
        return n.value

    def epoch_prefetch(self):
        _check(self._L.sbr_fit_epoch_prefetch(self._h))

    def minibatch_rows(self, mb: int) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_fit_minibatch_rows(self._h, mb, C.byref(n)))
        return n.value

    def step(self, mb: int):
        _check(self._L.sbr_fit_step(self._h, mb))

    def steps(self, first: int, count: int):
        """``count`` consecutive optimiser steps (sbr_fit_steps): one-sequence steps at d <= 32 run as one launch per run."""
        _check(self._L.sbr_fit_steps(self._h, first, count))

    def phase_clocks(self):
        """Per-phase ticks (100 MHz) of the one-launch step runs since fit_begin: forward, score + tail, backward, dense, sparse; steps."""
        out = (C.c_uint64 * 6)()
        _check(self._L.sbr_fit_debug_phase_clocks(self._h, out))
        return list(out)

    def step_local(self, mb: int):
        _check(self._L.sbr_fit_step_local(self._h, mb))

    def step_apply(self, mb: int):
        _check(self._L.sbr_fit_step_apply(self._h, mb))

    # ---- multi-device owner-reduce protocol (device pointers supplied by the caller) ----
    def chunk_bytes(self) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_fit_chunk_bytes(self._h, C.byref(n)))
        return n.value

    def dense_bytes(self) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_fit_dense_bytes(self._h, C.byref(n)))
        return n.value

    def step_scatter(self, mb: int, send_ptr: int):
        _check(self._L.sbr_fit_step_scatter(self._h, mb, C.c_void_p(send_ptr)))

    def step_dense(self, dense_ptr: int):
        _check(self._L.sbr_fit_step_dense(self._h, C.c_void_p(dense_ptr)))

    def step_owner_reduce(self, recv_ptr: int, own_ptr: int, stream_ptr: int = None):
        """``stream_ptr``: launch on that HIP stream without synchronising it or the model's stream (the caller
        orders them) — the exchange stream of the staleness-one pipeline."""
        if stream_ptr is None:
            _check(self._L.sbr_fit_step_owner_reduce(self._h, C.c_void_p(recv_ptr), C.c_void_p(own_ptr)))
        else:
            _check(self._L.sbr_fit_step_owner_reduce_on(self._h, C.c_void_p(recv_ptr), C.c_void_p(own_ptr), C.c_void_p(stream_ptr)))

    def step_apply_table(self, table_ptr: int, dense_all_ptr: int):
        _check(self._L.sbr_fit_step_apply_table(self._h, C.c_void_p(table_ptr), C.c_void_p(dense_all_ptr)))

    def step_apply_rows(self, table_ptr: int):
        """Item-table half of step_apply_table (opens the optimiser step; does not wait for the dense-gradient GEMM)."""
        _check(self._L.sbr_fit_step_apply_rows(self._h, C.c_void_p(table_ptr)))

    def step_apply_dense(self, dense_all_ptr: int):
        _check(self._L.sbr_fit_step_apply_dense(self._h, C.c_void_p(dense_all_ptr)))

    def step_owner_update(self, recv_ptr: int):
        """Owner-applied update (sbr_fit_step_owner_update): device-order sum of the devices' contributions to this rank's slice and
        the one optimiser update of its touched rows, in place; opens the optimiser step."""
        _check(self._L.sbr_fit_step_owner_update(self._h, C.c_void_p(recv_ptr)))

    def end(self):
        loss, ex = C.c_float(), C.c_uint64()
        _check(self._L.sbr_fit_end(self._h, C.byref(loss), C.byref(ex)))
        return loss.value, ex.value

    def end_lagged(self) -> float:
        """This device's term of the figure the reference's ``fit`` returns (stale loss-node values, sequence_model.rs:157)."""
        v = C.c_float()
        _check(self._L.sbr_fit_end_lagged(self._h, C.byref(v)))
        return v.value

    def counters(self):
        ex, neg = C.c_uint64(), C.c_uint64()
        _check(self._L.sbr_fit_counters(self._h, C.byref(ex), C.byref(neg)))
        return ex.value, neg.value

    def sparse_stats(self):
        """(gradient entries, distinct table rows) of the last step's sparse update on this device."""
        ent, uniq = C.c_uint64(), C.c_uint64()
        _check(self._L.sbr_fit_sparse_stats(self._h, C.byref(ent), C.byref(uniq)))
        return ent.value, uniq.value

    def debug_fetch(self, which: int, rows: int) -> np.ndarray:
        d = self.model.storage_dim
        which = int(which)
        if which in (0, 4, 5):
            out = np.zeros((rows, d), dtype=np.float32)
        elif which == 6:
            out = np.zeros(self.model.dense_count(), dtype=np.float32)
        elif which == 10:
            out = np.zeros((rows, {0: 4, 1: 3, 2: 0}[int(self.model.hp.model)] * d), dtype=np.float32)
        else:
            out = np.zeros(rows, dtype=np.uint32 if which in _DBG_U32 else np.float32)
        _check(self._L.sbr_fit_debug_fetch(self._h, which, _ptr(out), out.nbytes))
        return out

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._L.sbr_fit_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Owns one sbr_model handle (device-resident parameters)."""

    def __init__(self, hp: SbrHparams):
        self._L = _lib.load()
        self.hp = hp
        self.dim = int(hp.embedding_dim)
        h = C.c_void_p()
        _check(self._L.sbr_model_create(C.byref(hp), C.byref(h)))  # rejects an embedding_dim outside 1..256
        self._h = h
        self.storage_dim = storage_dim(self.dim)

    @classmethod
    def _from_handle(cls, hp: SbrHparams, handle) -> "Model":
        m = cls.__new__(cls)
        m._L = _lib.load()
        m.hp = hp
        m.dim = int(hp.embedding_dim)
        m.storage_dim = storage_dim(m.dim)
        m._h = handle
        return m

    def is_partitioned(self) -> bool:
        v = C.c_int32()
        _check(self._L.sbr_model_is_partitioned(self._h, C.byref(v)))
        return bool(v.value)

    def partition_parts(self):
        """[(home rank, bytes)] of the partitioned table's parts — runs of pages with one home, in address order over the arrays
        E, E_acc, (E_m), b, b_acc, (b_m) (sbr_partition_part_info)."""
        n = C.c_uint32()
        _check(self._L.sbr_partition_num_parts(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            home, nbytes = C.c_uint32(), C.c_uint64()
            _check(self._L.sbr_partition_part_info(self._h, i, C.byref(home), C.byref(nbytes)))
            out.append((home.value, nbytes.value))
        return out

    def dense_count(self) -> int:
        d = self.storage_dim
        ng = {0: 4, 1: 3, 2: 0}[int(self.hp.model)]
        return (2 * d + 1) * ng * d if ng else d

    def param_count(self, which: int) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_model_param_count(self._h, int(which), C.byref(n)))
        return n.value

    def get_param(self, which: int) -> np.ndarray:
        out = np.zeros(self.param_count(which), dtype=np.float32)
        if out.size:
            _check(self._L.sbr_model_get_param(self._h, int(which), _ptr(out), out.size))
        return out

    def get_param_rows(self, which: int, rows) -> np.ndarray:
        """Selected rows of an item-table block ([n, embedding_dim]; biases: [n]) without moving the whole table."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        table = self.param_count(which) != int(self.hp.num_items)
        out = np.zeros((rows.size, self.dim) if table else rows.size, dtype=np.float32)
        _check(self._L.sbr_model_get_param_rows(self._h, int(which), _ptr(rows), rows.size, _ptr(out)))
        return out

    def set_param(self, which: int, values):
        values = np.ascontiguousarray(values, dtype=np.float32).ravel()
        _check(self._L.sbr_model_set_param(self._h, int(which), _ptr(values), values.size))

    def global_epoch(self) -> int:
        n = C.c_uint64()
        _check(self._L.sbr_model_get_epoch(self._h, C.byref(n)))
        return n.value

    def counters(self):
        e, t = C.c_uint64(), C.c_uint64()
        _check(self._L.sbr_model_get_counters(self._h, C.byref(e), C.byref(t)))
        return e.value, t.value

    def set_counters(self, global_epoch: int, optimizer_steps: int):
        _check(self._L.sbr_model_set_counters(self._h, global_epoch, optimizer_steps))

    def get_rng(self) -> bytes:
        """The model RNG's state as the 16 bytes that re-create it through ``XorShiftRng.from_seed``."""
        buf = (C.c_uint8 * 16)()
        _check(self._L.sbr_model_get_rng(self._h, buf))
        return bytes(buf)

    def set_rng(self, state: bytes):
        state = bytes(state)
        if len(state) != 16:
            raise ValueError("RNG state is 16 bytes")
        _check(self._L.sbr_model_set_rng(self._h, (C.c_uint8 * 16)(*state)))

    def table_slice(self, which: int):
        """(device pointer of an item-table block of this replica, bytes per owner slice): rank r's slice is [ptr + r * bytes, + bytes)
        (sbr_model_table_slice; the block is allocated for num_devices equal slices).  (0, 0): the model has no such block."""
        base, nb = C.c_void_p(), C.c_uint64()
        _check(self._L.sbr_model_table_slice(self._h, int(which), C.byref(base), C.byref(nb)))
        return (base.value or 0), nb.value

    def optimizer_state_gathered(self):
        """The host has all-gathered the owners' optimiser-state slices into this replica (sbr_model_optimizer_state_gathered)."""
        _check(self._L.sbr_model_optimizer_state_gathered(self._h))

    def optimizer_state_is_partial(self) -> bool:
        v = C.c_int32()
        _check(self._L.sbr_model_optimizer_state_is_partial(self._h, C.byref(v)))
        return bool(v.value)

    def set_stream(self, hip_stream_ptr: int):
        _check(self._L.sbr_model_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def synchronize(self):
        _check(self._L.sbr_model_synchronize(self._h))

    def timing_enable(self, on: bool = True):
        _check(self._L.sbr_model_timing_enable(self._h, 1 if on else 0))

    def timing_select(self, families=None):
        """Which kernel families are bracketed by events while timing is on (names of ``KernelFamily``; None = all)."""
        mask = 0xFFFFFFFF if families is None else sum(1 << int(KernelFamily[f]) for f in families)
        _check(self._L.sbr_model_timing_select(self._h, mask))

    def set_reference_order(self, on: bool = True):
        """Negatives from the worker's own sequential xorshift stream (sequence_model.rs:58-65, :137) instead of the counter-keyed
        draws: one subsequence per step; one device, or a Synchronous replicated group driven by `group_fit` (every worker's gradient then
        goes in as its own optimiser step); max_sequence_length <= 256 / 221 / 81 at d <= 64 / 128 / 256 (sbr_model_set_reference_order)."""
        _check(self._L.sbr_model_set_reference_order(self._h, 1 if on else 0))

    def set_step_fusion(self, level: int):
        """How one-sequence steps at d <= 32 are launched: 0 separate launches, 1 fused launches (four per step), 2 (default)
        runs of steps in one launch where the shape allows (sbr_model_set_step_fusion).  Same bits."""
        _check(self._L.sbr_model_set_step_fusion(self._h, int(level)))

    def set_overlap(self, on: bool = True):
        """False: side-stream work runs on the main stream, so kernel families are timed standalone."""
        _check(self._L.sbr_model_set_overlap(self._h, 1 if on else 0))

    def timing_read(self):
        ms = (C.c_double * NUM_KERNEL_FAMILIES)()
        n = (C.c_uint64 * NUM_KERNEL_FAMILIES)()
        _check(self._L.sbr_model_timing_read(self._h, ms, n))
        return {KernelFamily(i).name: (ms[i], n[i]) for i in range(NUM_KERNEL_FAMILIES)}

    def fit(self, user_ptr, item_ids) -> float:
        up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        loss = C.c_float()
        _check(self._L.sbr_model_fit(self._h, _ptr(up), _ptr(it), len(up) - 1, C.byref(loss)))
        return loss.value

    def last_fit_lagged_loss(self) -> float:
        """What the reference's ``fit`` would have returned for the last ``fit`` / ``group_fit``; ``fit`` itself
        returns the true mean loss."""
        v = C.c_float()
        _check(self._L.sbr_model_last_fit_lagged_loss(self._h, C.byref(v)))
        return v.value

    def fit_begin(self, user_ptr, item_ids) -> FitPlan:
        return FitPlan(self, user_ptr, item_ids)

    def user_representation(self, item_ids) -> np.ndarray:
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        out = np.zeros(self.dim, dtype=np.float32)
        _check(self._L.sbr_user_representation(self._h, _ptr(it), it.size, _ptr(out)))
        return out

    def predict(self, user, item_ids) -> np.ndarray:
        user = np.ascontiguousarray(user, dtype=np.float32)
        if user.size != self.dim:
            raise ValueError("user representation has the wrong dimension")
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        out = np.zeros(it.size, dtype=np.float32)
        _check(self._L.sbr_predict(self._h, _ptr(user), _ptr(it), it.size, _ptr(out)))
        return out

    def mrr_score(self, user_ptr, item_ids):
        up = np.ascontiguousarray(user_ptr, dtype=np.uint64)
        it = np.ascontiguousarray(item_ids, dtype=np.uint32)
        ranks = np.zeros(max(len(up) - 1, 1), dtype=np.uint32)
        mrr, n = C.c_float(), C.c_uint64()
        _check(self._L.sbr_mrr_score(self._h, _ptr(up), _ptr(it), len(up) - 1, C.byref(mrr), _ptr(ranks), C.byref(n)))
        return mrr.value, ranks[: n.value].copy()

    def close(self):
        if getattr(self, "_h", None):
            self._L.sbr_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def release_cached_memory() -> None:
    """Hand the engine's idle fit scratch (device and pinned-host blocks kept between fit calls) back to the driver."""
    _lib.load().sbr_release_cached_memory()
