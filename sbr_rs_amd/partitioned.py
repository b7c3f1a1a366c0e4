"""Partitioned item table under one process per GPU (the launcher model of ``bench.py --gpus N``).

Every rank holds the rows ``[r*S, (r+1)*S)`` of the item table on its own device; the whole table is one
virtual address range in every process (HIP virtual memory management), the peers' parts being mapped
from file descriptors that travel over Unix sockets (``SCM_RIGHTS``).  The unchanged gather kernels read
remote rows through that mapping (xGMI); every row is updated by its owner from the ranks' gradient
lists, which are shared the same way.  Results are bit-identical to the single-process partitioned group
(``sbr_group_fit``) and to the replicated Synchronous exchange.

torch.distributed supplies the control plane only (rendezvous, two small all-gathers and one barrier per
step); the bulk data never goes through a collective.
"""
from __future__ import annotations

import array
import ctypes as C
import json
import os
import socket
import tempfile

import numpy as np

from . import _lib
from .engine import FitPlan, Model, _check


def _send_fds(path: str, payload: dict, fds) -> None:
    with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as s:
        s.connect(path)
        socket.send_fds(s, [json.dumps(payload).encode()], list(fds))


def _recv_fds(server: socket.socket, maxfds: int):
    conn, _ = server.accept()
    with conn:
        msg, fds, _flags, _addr = socket.recv_fds(conn, 1 << 16, maxfds)
        return json.loads(msg.decode()), list(fds)


class _FdExchange:
    """All-to-all of file descriptors between the ranks of one node."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        box = [tempfile.mkdtemp(prefix="sbr_fd_") if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        self.dir = box[0]
        self.server = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.server.bind(self._path(self.rank))
        self.server.listen(self.world)
        dist.barrier(group=group)

    def _path(self, r: int) -> str:
        return os.path.join(self.dir, f"r{r}.sock")

    def all_to_all(self, payload: dict, fds):
        """Sends (payload, fds) to every peer; returns {peer: (payload, fds)}."""
        for peer in range(self.world):
            if peer != self.rank:
                _send_fds(self._path(peer), dict(payload, src=self.rank), fds)
        out = {}
        for _ in range(self.world - 1):
            msg, got = _recv_fds(self.server, 64)
            out[int(msg["src"])] = (msg, got)
        self.dist.barrier(group=self.group)
        return out

    def close(self):
        self.server.close()
        self.dist.barrier(group=self.group)
        if self.rank == 0:
            for r in range(self.world):
                try:
                    os.unlink(self._path(r))
                except OSError:
                    pass
            try:
                os.rmdir(self.dir)
            except OSError:
                pass


def create_partitioned_model(hp, group=None) -> Model:
    """This rank's model over the partitioned table (hp.num_devices = world size, hp.device_rank = rank; the
    HIP device must already be selected)."""
    import torch.distributed as dist

    L = _lib.load()
    if int(hp.num_devices) != dist.get_world_size(group) or int(hp.device_rank) != dist.get_rank(group):
        raise RuntimeError("hp.num_devices / hp.device_rank must equal the process group's size / this rank")
    h = C.c_void_p()
    _check(L.sbr_model_create_partitioned(C.byref(hp), C.byref(h)))
    model = Model._from_handle(hp, h)
    nparts = C.c_uint32()
    _check(L.sbr_partition_num_parts(h, C.byref(nparts)))
    own, fds = [], []
    for i in range(nparts.value):
        home = C.c_uint32()
        _check(L.sbr_partition_part_info(h, i, C.byref(home), None))
        if home.value == int(hp.device_rank):
            fd = C.c_int32()
            _check(L.sbr_partition_export_part(h, i, C.byref(fd)))
            own.append(i)
            fds.append(fd.value)
    ex = _FdExchange(group)
    try:
        for _peer, (msg, got) in ex.all_to_all({"parts": own}, fds).items():
            for part, fd in zip(msg["parts"], got):
                _check(L.sbr_partition_import_part(h, int(part), int(fd)))
                os.close(fd)
    finally:
        ex.close()
    for fd in fds:
        os.close(fd)
    _check(L.sbr_partition_finalize(h))
    dist.barrier(group=group)  # every rank has written its rows before anybody reads the table
    return model


class PartitionedStepper:
    """The optimiser-step sequencing of one rank over a partitioned table (same surface as
    distributed.StepLoop: begin_epoch / step), used by fit_partitioned and bench.py."""

    def __init__(self, model: Model, interactions, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.L = _lib.load()
        self.model = model
        self.world = int(model.hp.num_devices)
        up, it = (interactions.user_pointers, interactions.item_ids) if hasattr(interactions, "user_pointers") else interactions
        model.set_stream(torch.cuda.current_stream().cuda_stream)
        self.plan: FitPlan = model.fit_begin(up, it)
        self.staged = dist.get_backend(group) == "gloo"  # gloo has no device collectives: small blocks go through the host
        fds4, bytes4 = (C.c_int32 * 4)(), (C.c_uint64 * 4)()
        _check(self.L.sbr_fit_lists_export(self.plan._h, fds4, bytes4))
        ex = _FdExchange(group)
        try:
            for peer, (msg, got) in ex.all_to_all({"bytes": [int(b) for b in bytes4]}, list(fds4)).items():
                _check(self.L.sbr_fit_lists_import(self.plan._h, peer, (C.c_int32 * 4)(*got), (C.c_uint64 * 4)(*msg["bytes"])))
                for fd in got:
                    os.close(fd)
        finally:
            ex.close()
        for fd in fds4:
            os.close(fd)
        self.db = self.plan.dense_bytes()
        self.dense = torch.zeros(self.db, dtype=torch.uint8, device="cuda")
        self.dense_all = torch.zeros(self.world * self.db, dtype=torch.uint8, device="cuda")
        self.bounds = np.zeros(self.world + 1, dtype=np.uint32)
        self.all_bounds = torch.zeros(self.world * (self.world + 1) * 4, dtype=torch.uint8, device="cuda")
        self.token = torch.zeros(1, dtype=torch.int32, device="cuda")
        self._bounds_view = None
        self.num_minibatches = 0

    def begin_epoch(self, prefetch_next: bool = False) -> int:
        self.num_minibatches = self.plan.epoch_prepare()
        if prefetch_next:
            self.plan.epoch_prefetch()
        return self.num_minibatches

    def _step_queued(self, mb: int) -> None:
        """Device collectives (RCCL): nothing of the step waits for the host.  The ranks' owner bounds and dense blocks are
        all-gathered ON THE DEVICE — a collective completes on a rank only after every rank's contribution, i.e. after every rank
        has finished READING the table — the owner builds its merge plan from the gathered bounds on the device, and a one-word
        all-reduce behind the owners' updates keeps the next step's reads behind every owner's WRITES."""
        from .distributed import device_bytes_as_tensor

        torch, dist, L, plan = self.torch, self.dist, self.L, self.plan
        plan.step_local(mb)
        bptr = C.c_void_p()
        _check(L.sbr_fit_step_reduce_own_queued(plan._h, mb, C.byref(bptr), C.c_void_p(self.dense.data_ptr())))
        if self._bounds_view is None or self._bounds_view[0] != bptr.value:
            self._bounds_view = (bptr.value, device_bytes_as_tensor(torch, bptr.value, (self.world + 1) * 4))
        dist.all_gather_into_tensor(self.all_bounds, self._bounds_view[1], group=self.group)
        dist.all_gather_into_tensor(self.dense_all, self.dense, group=self.group)
        _check(L.sbr_fit_step_owner_apply_queued(plan._h, C.c_void_p(self.all_bounds.data_ptr()), C.c_void_p(self.dense_all.data_ptr())))
        dist.all_reduce(self.token, group=self.group)  # every owner has finished WRITING its rows (ordered on the stream, not on the host)

    def step(self, mb: int) -> None:
        torch, dist, L, plan, world = self.torch, self.dist, self.L, self.plan, self.world
        if not self.staged:
            return self._step_queued(mb)
        plan.step_local(mb)
        _check(L.sbr_fit_step_reduce_own(plan._h, mb, self.bounds.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(self.dense.data_ptr())))
        # rendezvous: after these all-gathers every rank has finished READING the table
        tb = torch.from_numpy(self.bounds.astype(np.int64)).to("cpu" if self.staged else "cuda")
        gathered = [torch.zeros_like(tb) for _ in range(world)]
        dist.all_gather(gathered, tb, group=self.group)
        if self.staged:
            parts = [torch.empty(self.db, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, self.dense.cpu(), group=self.group)
            self.dense_all.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.dense_all, self.dense, group=self.group)
        all_bounds = np.ascontiguousarray(torch.stack([g.cpu() for g in gathered]).numpy().astype(np.uint32))
        torch.cuda.current_stream().synchronize()
        _check(L.sbr_fit_step_owner_apply(plan._h, all_bounds.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(self.dense_all.data_ptr())))
        dist.barrier(group=self.group)  # every owner has finished WRITING its rows

    def end(self):
        return self.plan.end()

    def close(self):
        self.plan.close()


def fit_partitioned(model: Model, interactions, group=None) -> float:
    """``fit`` over a partitioned table, one process per GPU (≙ fit with num_threads = world size)."""
    if int(model.hp.parallelism) == 0:
        import warnings

        warnings.warn("Parallelism.Asynchronous over a partitioned item table runs the synchronous step (its owners update in "
                      "place after a rendezvous; the result equals Parallelism.Synchronous, bit for bit)", stacklevel=2)
    stepper = PartitionedStepper(model, interactions, group)
    try:
        epochs = int(model.hp.num_epochs)
        for e in range(epochs):
            for mb in range(stepper.begin_epoch(prefetch_next=e + 1 < epochs)):
                stepper.step(mb)
        loss, _examples = stepper.end()
    finally:
        stepper.close()
    return loss


class PeerExchangeStepper:
    """The REPLICATED owner-reduce exchange with the peers' buffers read in place (peer transport): every rank
    exports its send buffer and its reduced own chunk once, the owner-reduce and table-update kernels read the
    peers' chunks through peer mappings (xGMI), and torch.distributed only orders the phases (two barriers and
    the all-gather of the small dense blocks per step).  Same surface as distributed.StepLoop, same bits as the
    collective transport."""

    def __init__(self, model: Model, interactions, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.L = _lib.load()
        self.model = model
        self.world = int(model.hp.num_devices)
        up, it = (interactions.user_pointers, interactions.item_ids) if hasattr(interactions, "user_pointers") else interactions
        model.set_stream(torch.cuda.current_stream().cuda_stream)
        self.plan: FitPlan = model.fit_begin(up, it)
        self.staged = dist.get_backend(group) == "gloo"
        fds2, bytes2 = (C.c_int32 * 2)(), (C.c_uint64 * 2)()
        _check(self.L.sbr_fit_exchange_export(self.plan._h, fds2, bytes2))
        ex = _FdExchange(group)
        try:
            for peer, (msg, got) in ex.all_to_all({"bytes": [int(b) for b in bytes2]}, list(fds2)).items():
                _check(self.L.sbr_fit_exchange_import(self.plan._h, peer, (C.c_int32 * 2)(*got), (C.c_uint64 * 2)(*msg["bytes"])))
                for fd in got:
                    os.close(fd)
        finally:
            ex.close()
        for fd in fds2:
            os.close(fd)
        self.db = self.plan.dense_bytes()
        self.dense = torch.zeros(self.db, dtype=torch.uint8, device="cuda")
        self.dense_all = torch.zeros(self.world * self.db, dtype=torch.uint8, device="cuda")
        self.num_minibatches = 0

    def begin_epoch(self, prefetch_next: bool = False) -> int:
        self.num_minibatches = self.plan.epoch_prepare()
        if prefetch_next:
            self.plan.epoch_prefetch()
        return self.num_minibatches

    def step(self, mb: int) -> None:
        torch, dist, L, plan = self.torch, self.dist, self.L, self.plan
        plan.step_local(mb)
        _check(L.sbr_fit_step_scatter_shared(plan._h, mb))       # this rank's send chunks are complete
        dist.barrier(group=self.group)                           # ... and so are everybody's
        _check(L.sbr_fit_step_owner_reduce_peers(plan._h))       # reads chunk `rank` of every peer's send buffer
        plan.step_dense(self.dense.data_ptr())
        self.model.synchronize()
        if self.staged:                                          # the all-gather doubles as "all own chunks are complete"
            parts = [torch.empty(self.db, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, self.dense.cpu(), group=self.group)
            self.dense_all.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.dense_all, self.dense, group=self.group)
        torch.cuda.current_stream().synchronize()
        _check(L.sbr_fit_step_apply_table_peers(plan._h, C.c_void_p(self.dense_all.data_ptr())))
        dist.barrier(group=self.group)                           # nobody is still reading this rank's buffers

    def end(self):
        return self.plan.end()

    def close(self):
        self.plan.close()
