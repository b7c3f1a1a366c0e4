"""`world` simulated ranks on ONE GPU through the multi-GPU protocol's C-ABI halves: everything is the production path
except the collectives, which are tensor copies here (include/sbr_hip.h "Multi-device step").  Two forms of the
Synchronous step (≙ sequence_model.rs:163-169 across devices), same bits:

  exchange = "owner"     scatter -> all-to-all -> sbr_fit_step_owner_update (the owner reduces AND updates its slice in place)
                         -> all-gather of the updated parameter slices into every replica's table -> dense
  exchange = "gradient"  scatter -> all-to-all -> sbr_fit_step_owner_reduce -> all-gather of the reduced gradient chunks ->
                         sbr_fit_step_apply_table on every replica (rounds 1-5; what the staleness-one pipeline still runs)
"""
from __future__ import annotations

import torch

from sbr_rs_amd._abi import Param
from sbr_rs_amd.distributed import device_bytes_as_tensor

PARAMETER_BLOCKS = (Param.ITEM_EMBEDDING, Param.ITEM_BIAS)
STATE_BLOCKS = (Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS_ACC, Param.ITEM_EMBEDDING_M, Param.ITEM_BIAS_M)


class SimulatedRanks:
    def __init__(self, models, plans):
        self.models, self.plans, self.world = models, plans, len(models)
        self.chunk, self.dbytes = plans[0].chunk_bytes(), plans[0].dense_bytes()
        u8 = dict(dtype=torch.uint8, device="cuda")
        w = self.world
        self.send = [torch.zeros(w * self.chunk, **u8) for _ in range(w)]
        self.dense = [torch.zeros(self.dbytes, **u8) for _ in range(w)]
        self.recv = torch.zeros(w * self.chunk, **u8)
        self.own = None
        self._views = {}

    def _block(self, q, which):
        """(replica q's whole block as a tensor aliasing the engine's memory, bytes per owner slice), or (None, 0)."""
        if (q, which) not in self._views:
            ptr, sb = self.models[q].table_slice(which)
            self._views[q, which] = (device_bytes_as_tensor(torch, ptr, sb * self.world), sb) if ptr else (None, 0)
        return self._views[q, which]

    def all_gather_slices(self, blocks):
        """slice p of every replica <- replica p's slice p (the in-place all-gather of sbr_model_table_slice)"""
        torch.cuda.synchronize()
        for which in blocks:
            for q in range(self.world):
                dst, sb = self._block(q, which)
                if dst is None:
                    continue
                for p in range(self.world):
                    if p != q:
                        src, _ = self._block(p, which)
                        dst[p * sb:(p + 1) * sb] = src[p * sb:(p + 1) * sb]
        torch.cuda.synchronize()

    def exchange(self, mb, exchange="owner", scatter=True):
        """The step's exchange + update after every rank's step_local(mb)."""
        w, chunk = self.world, self.chunk
        if scatter:
            for q in range(w):
                self.plans[q].step_scatter(mb, self.send[q].data_ptr())
                self.plans[q].step_dense(self.dense[q].data_ptr())
                self.models[q].synchronize()
        if exchange == "gradient" and self.own is None:
            self.own = [torch.zeros(chunk, dtype=torch.uint8, device="cuda") for _ in range(w)]
        for q in range(w):  # all_to_all_single
            for src in range(w):
                self.recv[src * chunk:(src + 1) * chunk] = self.send[src][q * chunk:(q + 1) * chunk]
            torch.cuda.synchronize()
            if exchange == "owner":
                self.plans[q].step_owner_update(self.recv.data_ptr())
            else:
                self.plans[q].step_owner_reduce(self.recv.data_ptr(), self.own[q].data_ptr())
            self.models[q].synchronize()
        dense_all = torch.cat(self.dense)
        if exchange == "owner":
            self.all_gather_slices(PARAMETER_BLOCKS)
            for q in range(w):
                self.plans[q].step_apply_dense(dense_all.data_ptr())
                self.models[q].synchronize()
        else:
            table = torch.cat(self.own)  # all_gather_into_tensor
            torch.cuda.synchronize()
            for q in range(w):
                self.plans[q].step_apply_table(table.data_ptr(), dense_all.data_ptr())
                self.models[q].synchronize()

    def step(self, mb, exchange="owner"):
        for q in range(self.world):
            self.plans[q].step_local(mb)
        self.exchange(mb, exchange)

    def finish(self):
        """End of a fit: the owners' optimiser-state slices back on every replica (a no-op after gradient-form steps)."""
        if any(m.optimizer_state_is_partial() for m in self.models):
            self.all_gather_slices(STATE_BLOCKS)
            for m in self.models:
                m.optimizer_state_gathered()
