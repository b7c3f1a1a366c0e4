"""User-sharded multi-GPU training: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI) for the exchange, the C-ABI for everything else.

What the reference does (sequence_model.rs:90-102, 163-169; mod.rs:35-41): the shuffled
subsequences are cut into ``num_threads`` partitions, every thread works on its own partition
against parameters shared through memory, and in ``Parallelism::Synchronous`` the threads
rendezvous at every optimiser step.  Here a partition lives on a device, the "shared memory" is a
full parameter replica per GPU, and the rendezvous is the **owner-reduce** exchange:

    compute_local   forward / negative sampling / backward on the device's own minibatch
    scatter         the device's own sparse entries reduced per table row into ``world`` dense
                    chunks, one per owner of a contiguous slice of ceil(I / world) rows
    all_to_all      chunk p of every device -> device p                      (RCCL, xGMI mesh)
    owner_update    the owner adds the devices' contributions in device order AND applies the one
                    optimiser update of every touched row of its slice, in place (round 6; ≙ the
                    one shared parameter + optimiser state of lstm.rs:259-260)
    all_gather      of the owners' updated PARAMETER slices, in place into every replica's table,
                    and of the small dense blocks (LSTM / alpha gradients + loss header)  (RCCL)
    apply_dense     every device applies the identical dense update (device-order sum)

(Parallelism::Asynchronous, the staleness-one pipeline, keeps rounds 1-5's form of the last three:
owner_reduce -> all-gather of the reduced GRADIENT chunks -> apply_table on every replica.)

Per device and step the exchange moves 2 x (world-1)/world x table bytes, independent of the
batch, and every pairwise link of the xGMI mesh carries table/world bytes per phase — instead of
(world-1) x (2·4d + 16) bytes per interaction for an all-gather of the raw entries.  Every
reduction has a fixed order, so replicas stay bit-identical without any parameter broadcast, and
the result equals the single-process oracle run with ``num_devices = world`` bit for bit.  During a
fit a row's optimiser state is maintained by its owner alone; when the fit ends the owners' state
slices are all-gathered the same way, so every replica is complete again (checkpoints, get_param).

The driver below is backend-agnostic on purpose: ``tests/test_distributed_cpu.py`` runs it with
world_size 2 on the gloo backend with CPU tensors (the oracle computing the device halves), which
covers the sharding and exchange logic without a GPU; ``tests/test_parity_gpu.py`` drives the HIP
halves for several simulated ranks on one GPU.
"""
from __future__ import annotations

import os
from typing import Protocol


class StepBackend(Protocol):
    """One device's half-steps around the exchange; tensors are 1-D uint8 on the exchange device."""

    def epoch_prepare(self, prefetch_next: bool = False) -> int: ...
    def compute_local(self, minibatch: int) -> None: ...
    def apply_single(self, minibatch: int) -> None: ...            # world == 1
    def scatter(self, minibatch: int): ...                         # -> send [world*chunk]
    def dense(self): ...                                           # -> dense block [dense_bytes]
    def owner_reduce(self, recv): ...                              # -> own chunk [chunk]
    def apply_table(self, table, dense_all) -> None: ...
    # optional: apply_rows(table) + apply_dense(dense_all) = apply_table in two halves (rows first)
    # optional, the owner-applied form: owner_update(recv) -> [(whole block, this rank's slice of it), ...] to all-gather
    #           (the HIP engine's pairs alias its table: the gather lands in place), slices_gathered(), apply_dense(dense_all);
    #           optimizer_state_slices() -> the same pairs for the optimiser-state blocks, optimizer_state_gathered()
    def buffers(self, world: int): ...                             # -> (recv, table, dense_all)
    def end(self): ...                                             # -> (loss, examples)


def _all_gather_pairs(dist, group, pairs, staged: bool) -> None:
    import torch

    for full, mine in pairs:
        if staged:  # gloo with device tensors: through host memory
            parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(dist.get_world_size(group))]
            dist.all_gather(parts, mine.cpu(), group=group)
            full.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(full, mine, group=group)


def owner_exchange_step(backend: StepBackend, minibatch: int, world: int, bufs, group=None) -> None:
    """One Synchronous optimiser step after compute_local, owner-applied: all-to-all of the gradient chunks, the owner's
    in-place update, all-gather of the updated parameter slices into every replica's table."""
    import torch
    import torch.distributed as dist

    recv, _table, dense_all = bufs
    send = backend.scatter(minibatch)
    staged = send.is_cuda and dist.get_backend(group) == "gloo"
    if staged:
        h_recv = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(h_recv, send.cpu(), group=group)
        recv.copy_(h_recv)
    else:
        dist.all_to_all_single(recv, send, group=group)
    _all_gather_pairs(dist, group, backend.owner_update(recv), staged)
    backend.slices_gathered()
    dense = backend.dense()  # joins the dense-gradient GEMM, which ran beside all of the above
    _all_gather_pairs(dist, group, [(dense_all, dense)], staged)
    backend.apply_dense(dense_all)


def gather_optimizer_state(backend: StepBackend, group=None) -> None:
    """After the last owner-applied step of a fit: every replica's copy of the item table's optimiser state made complete."""
    import torch.distributed as dist

    pairs = backend.optimizer_state_slices()
    staged = bool(pairs) and pairs[0][0].is_cuda and dist.get_backend(group) == "gloo"
    _all_gather_pairs(dist, group, pairs, staged)
    backend.optimizer_state_gathered()


def exchange_step(backend: StepBackend, minibatch: int, world: int, bufs, group=None) -> None:
    """One optimiser step after compute_local with the GRADIENT all-gather (rounds 1-5; what the pipeline's ordering is built
    on): owner reduce, all-gather of the reduced chunks, every replica applies every update."""
    import torch.distributed as dist

    recv, table, dense_all = bufs
    send = backend.scatter(minibatch)
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device collectives: stage through host memory.  This is the transport for hosts
        # without RCCL peers (two ranks sharing one GPU in tests/test_distributed_gpu.py); "nccl" is
        # the production transport and keeps everything on the device.
        _staged_exchange(backend, dist, group, send, recv, table, dense_all)
        return
    # the dense-gradient GEMM (the engine's side stream) runs beside the whole sparse exchange — scatter, all-to-all,
    # owner reduction and the all-gather of the reduced chunks are all enqueued before dense() joins it
    dist.all_to_all_single(recv, send, group=group)
    own = backend.owner_reduce(recv)
    dist.all_gather_into_tensor(table, own, group=group)
    apply_rows = getattr(backend, "apply_rows", None)
    if apply_rows is not None:
        apply_rows(table)  # the item-table update does not need the dense gradient either
    dense = backend.dense()
    dist.all_gather_into_tensor(dense_all, dense, group=group)
    if apply_rows is not None:
        backend.apply_dense(dense_all)
    else:
        backend.apply_table(table, dense_all)


def _staged_exchange(backend, dist, group, send, recv, table, dense_all, dense=None) -> None:
    import torch

    world = dist.get_world_size(group)
    h_recv = torch.empty(recv.shape, dtype=recv.dtype)
    dist.all_to_all_single(h_recv, send.cpu(), group=group)
    recv.copy_(h_recv)
    own = backend.owner_reduce(recv)
    for dst, src in ((table, own), (dense_all, backend.dense() if dense is None else dense)):
        parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
        dist.all_gather(parts, src.cpu(), group=group)
        dst.copy_(torch.cat(parts))
    backend.apply_table(table, dense_all)


class StepLoop:
    """The optimiser-step sequencing of one device, shared by ``run_fit`` and ``bench.py``.

    Synchronous (``Parallelism::Synchronous``, mod.rs:39-40): compute, exchange, apply.

    Asynchronous (``Parallelism::Asynchronous``, mod.rs:36-38) with more than one device: the
    deterministic analogue of Hogwild, staleness fixed at one step — minibatch k+1 is computed on
    parameters that lack update k, so the exchange of step k (on its own stream, when the backend
    has streams) runs underneath that computation; update k is applied afterwards.  With one
    device both modes are the same step."""

    def __init__(self, backend: StepBackend, world: int, asynchronous: bool = False, group=None, exchange: str = None):
        self.backend, self.world, self.group = backend, world, group
        self.asynchronous = bool(asynchronous) and world > 1
        # Synchronous: the owner-applied update where the backend has it; exchange="gradient" forces rounds 1-5's form (A/B, parity)
        want = exchange or os.environ.get("SBR_EXCHANGE", "owner")
        self.owner_applied = (not self.asynchronous) and world > 1 and want != "gradient" and hasattr(backend, "owner_update")
        self.bufs = (backend.buffers(world, gradient_gather=not self.owner_applied) if self.owner_applied else backend.buffers(world)) if world > 1 else None
        self.num_minibatches = 0
        self._computed = -1  # minibatch whose local results are in the backend's block
        self._owner_steps = 0

    def begin_epoch(self, prefetch_next: bool = False) -> int:
        self.num_minibatches = self.backend.epoch_prepare(prefetch_next=prefetch_next)
        self._computed = -1
        return self.num_minibatches

    def step(self, minibatch: int) -> None:
        be = self.backend
        if self._computed != minibatch:
            be.compute_local(minibatch)
        if self.world == 1:
            be.apply_single(minibatch)
        elif self.owner_applied:
            owner_exchange_step(be, minibatch, self.world, self.bufs, self.group)
            self._owner_steps += 1
        elif not self.asynchronous:
            exchange_step(be, minibatch, self.world, self.bufs, self.group)
        else:
            nxt = minibatch + 1 if minibatch + 1 < self.num_minibatches else None
            pipelined_exchange_step(be, minibatch, nxt, self.bufs, self.group)
            if nxt is not None:
                self._computed = nxt

    def finish(self) -> None:
        """End of a fit: the owners' optimiser-state slices back on every replica (no-op for the other step forms)."""
        if self.owner_applied and self._owner_steps:
            gather_optimizer_state(self.backend, self.group)
            self._owner_steps = 0


def pipelined_exchange_step(backend: StepBackend, minibatch: int, next_minibatch, bufs, group=None) -> None:
    """Asynchronous step: [scatter, dense] -> {exchange on the side || compute_local(next)} -> apply."""
    import torch.distributed as dist

    recv, table, dense_all = bufs
    send = backend.scatter(minibatch)
    dense = backend.dense()
    streams = getattr(backend, "exchange_streams", None)
    if streams is None or dist.get_backend(group) == "gloo":
        # no device streams to overlap on (CPU backend, or host-staged gloo): same order of effects
        if next_minibatch is not None:
            backend.compute_local(next_minibatch)
        if send.is_cuda:
            _staged_exchange(backend, dist, group, send, recv, table, dense_all, dense=dense)
        else:
            dist.all_to_all_single(recv, send, group=group)
            own = backend.owner_reduce(recv)
            dist.all_gather_into_tensor(table, own, group=group)
            dist.all_gather_into_tensor(dense_all, dense, group=group)
            backend.apply_table(table, dense_all)
        return
    with streams() as (torch, compute, side):
        side.wait_stream(compute)                      # send / dense are complete
        with torch.cuda.stream(side):
            dist.all_to_all_single(recv, send, group=group)
            own = backend.owner_reduce(recv, stream=side)
            dist.all_gather_into_tensor(table, own, group=group)
            dist.all_gather_into_tensor(dense_all, dense, group=group)
        if next_minibatch is not None:
            backend.compute_local(next_minibatch)      # on the compute stream, under the exchange
        compute.wait_stream(side)
    backend.apply_table(table, dense_all)


def run_fit(backend: StepBackend, num_epochs: int, world: int, group=None, asynchronous: bool = False, exchange: str = None):
    """The epoch/minibatch loop of fit_sequence_model (sequence_model.rs:108-171) with the
    per-step rendezvous expressed as collectives."""
    loop = StepLoop(backend, world, asynchronous, group, exchange=exchange)
    for e in range(num_epochs):
        nmb = loop.begin_epoch(prefetch_next=e + 1 < num_epochs)
        for mb in range(nmb):
            loop.step(mb)
    loop.finish()
    return backend.end()


class _DeviceBytes:
    """Device memory owned by the engine, described through ``__cuda_array_interface__`` so that torch wraps it without a copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}


def device_bytes_as_tensor(torch, ptr: int, nbytes: int):
    """A uint8 CUDA tensor ALIASING [ptr, ptr + nbytes) of the engine's memory (the model keeps it alive)."""
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")


class HipBackend:
    """The gfx950 engine as a StepBackend: exchange buffers are torch CUDA tensors whose device
    pointers go straight into the C-ABI; the engine is put on torch's current stream so no extra
    synchronisation is needed around the collectives."""

    def __init__(self, model, interactions_or_csr, world: int = 1):
        import torch

        self.torch = torch
        self.model = model
        self.world = world
        if hasattr(interactions_or_csr, "user_pointers"):
            up, it = interactions_or_csr.user_pointers, interactions_or_csr.item_ids
        else:
            up, it = interactions_or_csr
        self._compute = torch.cuda.current_stream()
        self._side = torch.cuda.Stream() if world > 1 else None
        model.set_stream(self._compute.cuda_stream)
        self.plan = model.fit_begin(up, it)
        self._views = {}
        if world > 1:
            self.chunk = self.plan.chunk_bytes()
            self.dense_bytes = self.plan.dense_bytes()
            self.send = torch.zeros(world * self.chunk, dtype=torch.uint8, device="cuda")
            self._dense = torch.zeros(self.dense_bytes, dtype=torch.uint8, device="cuda")
            self.own = torch.zeros(self.chunk, dtype=torch.uint8, device="cuda")

    def epoch_prepare(self, prefetch_next: bool = False) -> int:
        n = self.plan.epoch_prepare()
        if prefetch_next:
            self.plan.epoch_prefetch()
        return n

    def compute_local(self, minibatch: int) -> None:
        self.plan.step_local(minibatch)

    def apply_single(self, minibatch: int) -> None:
        self.plan.step_apply(minibatch)

    def scatter(self, minibatch: int):
        self.plan.step_scatter(minibatch, self.send.data_ptr())
        return self.send

    def dense(self):
        self.plan.step_dense(self._dense.data_ptr())
        return self._dense

    def owner_reduce(self, recv, stream=None):
        # stream != None: launched on the exchange stream (Asynchronous) with no host-side synchronisation of either
        # stream, so that compute_local(k+1) is enqueued while the exchange of step k is still in flight
        self.plan.step_owner_reduce(recv.data_ptr(), self.own.data_ptr(), None if stream is None else stream.cuda_stream)
        return self.own

    def exchange_streams(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            yield self.torch, self._compute, self._side

        return ctx()

    def apply_table(self, table, dense_all) -> None:
        self.plan.step_apply_table(table.data_ptr(), dense_all.data_ptr())

    def apply_rows(self, table) -> None:
        self.plan.step_apply_rows(table.data_ptr())

    def apply_dense(self, dense_all) -> None:
        self.plan.step_apply_dense(dense_all.data_ptr())

    # ---- owner-applied form: the engine's table blocks as torch tensors (no copy), this rank's slice a view of each ----
    def _block_pairs(self, blocks):
        rank = int(self.model.hp.device_rank)
        pairs = []
        for which in blocks:
            if which not in self._views:
                ptr, sb = self.model.table_slice(which)
                self._views[which] = device_bytes_as_tensor(self.torch, ptr, sb * self.world) if ptr else None
            full = self._views[which]
            if full is not None:
                sb = full.numel() // self.world
                pairs.append((full, full[rank * sb:(rank + 1) * sb]))
        return pairs

    def owner_update(self, recv):
        from ._abi import Param

        self.plan.step_owner_update(recv.data_ptr())
        return self._block_pairs((Param.ITEM_EMBEDDING, Param.ITEM_BIAS))

    def slices_gathered(self) -> None:
        pass  # the all-gather wrote the table in place

    def optimizer_state_slices(self):
        from ._abi import Param

        return self._block_pairs((Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS_ACC, Param.ITEM_EMBEDDING_M, Param.ITEM_BIAS_M))

    def optimizer_state_gathered(self) -> None:
        self.torch.cuda.current_stream().synchronize()
        self.model.optimizer_state_gathered()

    def buffers(self, world: int, gradient_gather: bool = True):
        t = self.torch
        return (t.zeros(world * self.chunk, dtype=t.uint8, device="cuda"),
                t.zeros(world * self.chunk, dtype=t.uint8, device="cuda") if gradient_gather else None,
                t.zeros(world * self.dense_bytes, dtype=t.uint8, device="cuda"))

    def end(self):
        return self.plan.end()

    def close(self):
        self.plan.close()


def fit_distributed(model, interactions, group=None, transport: str = None) -> float:
    """``fit`` for ``num_threads`` (= world size) > 1.  Requires an initialised process group whose
    size equals ``hp.num_devices`` and whose rank equals ``hp.device_rank``."""
    import torch.distributed as dist

    world = int(model.hp.num_devices)
    if not dist.is_initialized() or dist.get_world_size(group) != world:
        raise RuntimeError(f"fit with num_threads={world} needs torch.distributed initialised with world size {world}")
    if dist.get_rank(group) != int(model.hp.device_rank):
        raise RuntimeError("process rank does not match hp.device_rank")
    import os

    transport = transport or os.environ.get("SBR_EXCHANGE_TRANSPORT", "collective")
    if transport == "peer":  # chunks read in place through peer mappings; collectives only order the phases
        from .partitioned import PeerExchangeStepper

        if int(model.hp.parallelism) == 0:  # Parallelism::Asynchronous: only the collective transport has the pipelined order
            import warnings

            warnings.warn("Parallelism.Asynchronous with the peer transport runs the synchronous step (the staleness-one "
                          "pipeline belongs to the collective transport); the result equals Parallelism.Synchronous", stacklevel=2)

        stepper = PeerExchangeStepper(model, interactions, group)
        try:
            epochs = int(model.hp.num_epochs)
            for e in range(epochs):
                for mb in range(stepper.begin_epoch(prefetch_next=e + 1 < epochs)):
                    stepper.step(mb)
            loss, _examples = stepper.end()
        finally:
            stepper.close()
        return loss
    backend = HipBackend(model, interactions, world)
    try:
        loss, _examples = run_fit(backend, int(model.hp.num_epochs), world, group,
                                  asynchronous=int(model.hp.parallelism) == 0)
    finally:
        backend.close()
    return loss
