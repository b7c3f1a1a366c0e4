"""User-sharded multi-GPU training: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI) for the exchange, the C-ABI for everything else.

What the reference does (sequence_model.rs:90-102, 163-169; mod.rs:35-41): the shuffled
subsequences are cut into ``num_threads`` partitions, every thread works on its own partition
against parameters shared through memory, and in ``Parallelism::Synchronous`` the threads
rendezvous at every optimiser step.  Here a partition lives on a device, the "shared memory" is a
full parameter replica per GPU, and the rendezvous is one all-gather per step of the devices'
*exchange blocks* (packed rows' indices, loss coefficients, hidden states, input gradients and
the dense gradient block — layout in DESIGN.md §5).  Every device then applies the identical,
deterministically ordered update (dense: device-order sum; sparse: sorted by (row, device, packed
row)), so replicas stay bit-identical without any parameter broadcast, and the result equals the
single-process oracle run with ``num_devices = world`` bit for bit.

The driver below is backend-agnostic on purpose: ``tests/test_distributed_cpu.py`` runs it with
world_size 2 on the gloo backend with CPU tensors (the oracle computing the local halves), which
covers the sharding and exchange logic without a GPU.
"""
from __future__ import annotations

from typing import Protocol

import numpy as np


class StepBackend(Protocol):
    """One device's half-steps around the exchange."""

    def epoch_prepare(self) -> int: ...
    def local_block(self, minibatch: int): ...          # -> 1-D uint8 tensor on the exchange device
    def gathered_buffer(self, world: int): ...          # -> 1-D uint8 tensor [world * block_bytes]
    def apply(self, minibatch: int, gathered) -> None: ...
    def end(self): ...                                  # -> (loss, examples)


def run_fit(backend: StepBackend, num_epochs: int, world: int, group=None):
    """The epoch/minibatch loop of fit_sequence_model (sequence_model.rs:108-171) with the
    per-step rendezvous expressed as an all-gather."""
    import torch.distributed as dist

    gathered = backend.gathered_buffer(world)
    for e in range(num_epochs):
        try:
            nmb = backend.epoch_prepare(prefetch_next=e + 1 < num_epochs)
        except TypeError:  # backends without host-side prefetch
            nmb = backend.epoch_prepare()
        for mb in range(nmb):
            local = backend.local_block(mb)
            if world > 1:
                dist.all_gather_into_tensor(gathered, local, group=group)
                backend.apply(mb, gathered)
            else:
                backend.apply(mb, local)
    return backend.end()


class HipBackend:
    """The gfx950 engine as a StepBackend: exchange blocks are torch CUDA tensors whose device
    pointers go straight into sbr_fit_step_local / sbr_fit_step_apply; the engine is put on
    torch's current stream so no extra synchronisation is needed around the collective."""

    def __init__(self, model, interactions_or_csr):
        import torch

        self.torch = torch
        self.model = model
        if hasattr(interactions_or_csr, "user_pointers"):
            up, it = interactions_or_csr.user_pointers, interactions_or_csr.item_ids
        else:
            up, it = interactions_or_csr
        model.set_stream(torch.cuda.current_stream().cuda_stream)
        self.plan = model.fit_begin(up, it)
        self.block_bytes = self.plan.exchange_bytes()
        self.local = torch.zeros(self.block_bytes, dtype=torch.uint8, device="cuda")

    def epoch_prepare(self, prefetch_next: bool = False) -> int:
        n = self.plan.epoch_prepare()
        if prefetch_next:
            self.plan.epoch_prefetch()
        return n

    def local_block(self, minibatch: int):
        self.plan.step_local(minibatch, self.local.data_ptr())
        return self.local

    def gathered_buffer(self, world: int):
        return self.torch.zeros(world * self.block_bytes, dtype=self.torch.uint8, device="cuda")

    def apply(self, minibatch: int, gathered) -> None:
        self.plan.step_apply(minibatch, gathered.data_ptr())

    def end(self):
        return self.plan.end()

    def close(self):
        self.plan.close()


def fit_distributed(model, interactions, group=None) -> float:
    """``fit`` for ``num_threads`` (= world size) > 1.  Requires an initialised process group whose
    size equals ``hp.num_devices`` and whose rank equals ``hp.device_rank``."""
    import torch.distributed as dist

    world = int(model.hp.num_devices)
    if not dist.is_initialized() or dist.get_world_size(group) != world:
        raise RuntimeError(f"fit with num_threads={world} needs torch.distributed initialised with world size {world}")
    if dist.get_rank(group) != int(model.hp.device_rank):
        raise RuntimeError("process rank does not match hp.device_rank")
    backend = HipBackend(model, interactions)
    try:
        loss, _examples = run_fit(backend, int(model.hp.num_epochs), world, group)
    finally:
        backend.close()
    return loss
