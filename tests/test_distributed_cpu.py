"""World-size-2 test of the multi-device path on CPU (gloo): the distributed driver
(sbr_rs_amd/distributed.run_fit: all-to-all + all-gather owner-reduce exchange) is run by two
processes, the oracle computing each device's halves; the replicas must end bit-identical to each other and to the single-process
oracle emulating both devices (≙ Parallelism::Synchronous with num_threads = 2,
/root/reference/src/models/sequence_model.rs:91-98, 163-166; reference test lstm.rs:474-496)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import LOSS_HINGE, LOSS_WARP, PAR_ASYNC, PAR_SYNC, hparams, synthetic_interactions
from oracle.oracle import OracleModel
from sbr_rs_amd._abi import ModelKind, Param
from sbr_rs_amd.distributed import run_fit


class OracleBackend:
    """The oracle's device halves behind the same StepBackend interface the HIP engine uses."""

    def __init__(self, model: OracleModel, rank: int, world: int, ptr, items):
        self.rank, self.world, self.model = rank, world, model
        self.plan = model.fit_begin(ptr, items)

    def epoch_prepare(self, prefetch_next=False):
        return self.plan.epoch_prepare()

    def compute_local(self, mb):
        self.plan.compute_local(mb, self.rank)

    def apply_single(self, mb):
        self.plan.step_apply(self.plan.step_local(mb, device=0))

    def scatter(self, mb):
        return torch.from_numpy(self.plan.scatter(self.rank, self.world))

    def dense(self):
        return torch.from_numpy(self.plan.export_dense(self.rank))

    def owner_reduce(self, recv):
        return torch.from_numpy(self.plan.owner_reduce(recv.numpy()))

    def apply_table(self, table, dense_all):
        self.plan.apply_table(table.numpy(), dense_all.numpy())

    # ---- the owner-applied form (sbr_fit_step_owner_update): the checker's replicas hold plain arrays, so the gathered
    # slices are installed by slices_gathered() where the engine's all-gather lands in its table directly ----
    def _pairs(self, blocks):
        self._gathering = []
        for which in blocks:
            if self.model.param_count(which) == 0:
                continue
            mine = torch.from_numpy(self.model.table_slice(which, self.rank))
            full = torch.zeros(self.world * mine.numel(), dtype=torch.uint8)
            self._gathering.append((which, full))
        return [(full, torch.from_numpy(self.model.table_slice(which, self.rank))) for which, full in self._gathering]

    def _install(self):
        for which, full in self._gathering:
            sb = full.numel() // self.world
            for r in range(self.world):
                if r != self.rank:
                    self.model.set_table_slice(which, r, full[r * sb:(r + 1) * sb].numpy())
        self._gathering = []

    def owner_update(self, recv):
        self.plan.owner_update(self.rank, recv.numpy())
        return self._pairs((Param.ITEM_EMBEDDING, Param.ITEM_BIAS))

    def slices_gathered(self):
        self._install()

    def apply_dense(self, dense_all):
        self.plan.apply_dense_blocks(dense_all.numpy())

    def optimizer_state_slices(self):
        return self._pairs((Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS_ACC, Param.ITEM_EMBEDDING_M, Param.ITEM_BIAS_M))

    def optimizer_state_gathered(self):
        self._install()

    def buffers(self, world, gradient_gather=True):
        c, d = self.plan.chunk_bytes(), self.plan.dense_bytes()
        return (torch.zeros(world * c, dtype=torch.uint8), torch.zeros(world * c, dtype=torch.uint8) if gradient_gather else None,
                torch.zeros(world * d, dtype=torch.uint8))

    def end(self):
        return self.plan.end()


PARAMS = {2: [Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.ITEM_BIAS_ACC, Param.EWMA_ALPHA],
          0: [Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.LSTM_W, Param.LSTM_W_ACC, Param.LSTM_B]}


def _worker(rank, world, port, kind, loss, par, out_dir, exchange="owner"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ptr, items = synthetic_interactions(40 if world < 8 else 120, 90, 14, seed=5, zipf=True)
        hp = hparams(90, 10, 16, kind, loss, epochs=2, B=4, ndev=world, rank=rank, par=par)
        m = OracleModel(hp)
        loss_v, ex = run_fit(OracleBackend(m, rank, world, ptr, items), 2, world, asynchronous=par == PAR_ASYNC, exchange=exchange)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss_v, ex=ex,
                 **{p.name: m.get_param(p) for p in PARAMS[kind]})
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind,loss,par,world,exchange", [
    (int(ModelKind.EWMA), LOSS_WARP, PAR_SYNC, 2, "owner"), (int(ModelKind.LSTM_NORMAL), LOSS_HINGE, PAR_SYNC, 2, "owner"),
    (int(ModelKind.LSTM_NORMAL), LOSS_HINGE, PAR_SYNC, 2, "gradient"),   # rounds 1-5's Synchronous step: gradient all-gather, every replica applies
    (int(ModelKind.LSTM_NORMAL), LOSS_WARP, PAR_ASYNC, 2, "owner"),      # (the pipeline always runs the gradient all-gather)
    (int(ModelKind.LSTM_NORMAL), LOSS_WARP, PAR_SYNC, 8, "owner"), (int(ModelKind.EWMA), LOSS_HINGE, PAR_SYNC, 3, "owner"),
    (int(ModelKind.EWMA), LOSS_HINGE, PAR_ASYNC, 8, "owner")])
def test_multi_process_gloo_matches_single_process(tmp_path, oracle_lib, kind, loss, par, world, exchange):
    """`world` gloo processes through the production driver (world 8 = the node BASELINE configs[3] / [4] name) against one
    process that emulates all devices.  Synchronous = the owner-applied update (the owner of a slice reduces AND updates it; the
    updated parameter slices are all-gathered, the optimiser-state slices when the fit ends; world 3 over 90 items: ragged last
    slice) or, exchange = "gradient", the gradient all-gather of rounds 1-5 — both must equal the one-process emulation, whose
    step is the latter.  par = Asynchronous: the driver's pipelined step (compute k+1 before update k lands) must equal the
    oracle's staleness-one emulation."""
    mp.spawn(_worker, args=(world, _free_port(), kind, loss, par, str(tmp_path), exchange), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # single process, all devices emulated
    ptr, items = synthetic_interactions(40 if world < 8 else 120, 90, 14, seed=5, zipf=True)
    hp = hparams(90, 10, 16, kind, loss, epochs=2, B=4, ndev=world, rank=0, par=par)
    ref = OracleModel(hp)
    ref_loss = ref.fit(ptr, items)
    for p in PARAMS[kind]:
        c = ref.get_param(p)
        for r in range(world):
            assert np.array_equal(ranks[r][p.name].view(np.uint32), c.view(np.uint32)), f"rank {r} of {world} != single-process on {p.name}"
    assert all(float(r["loss"]) == float(ranks[0]["loss"]) for r in ranks) and float(ranks[0]["loss"]) == pytest.approx(ref_loss, rel=1e-6)
    assert all(int(r["ex"]) == int(ranks[0]["ex"]) for r in ranks) and int(ranks[0]["ex"]) > 0


def test_asynchronous_differs_from_synchronous_only_with_peers(oracle_lib):
    """Staleness exists only between workers: one device trains identically in both modes, two do not."""
    ptr, items = synthetic_interactions(40, 90, 14, seed=5, zipf=True)
    out = {}
    for ndev in (1, 2):
        for par in (PAR_SYNC, PAR_ASYNC):
            m = OracleModel(hparams(90, 10, 16, int(ModelKind.LSTM_NORMAL), LOSS_HINGE, epochs=2, B=4, ndev=ndev, par=par))
            m.fit(ptr, items)
            out[ndev, par] = m.get_param(Param.ITEM_EMBEDDING)
    assert np.array_equal(out[1, PAR_SYNC], out[1, PAR_ASYNC])
    assert not np.array_equal(out[2, PAR_SYNC], out[2, PAR_ASYNC])


def test_partitioning_drops_remainder_and_shards_disjointly(oracle_lib):
    """sequence_model.rs:91-98: len / num_threads per partition, remainder never trained."""
    ptr, items = synthetic_interactions(41, 60, 9, seed=2)
    nseq1 = None
    rows = {}
    for ndev in (1, 2, 3):
        hp = hparams(60, 9, 16, int(ModelKind.EWMA), LOSS_HINGE, epochs=1, B=1000, ndev=ndev)
        m = OracleModel(hp)
        plan = m.fit_begin(ptr, items)
        assert plan.epoch_prepare() == 1
        rows[ndev] = [plan.minibatch_rows(0, device=q) for q in range(ndev)]
    # B=1000 => one minibatch per device holding its whole partition; partitions are equal-sized
    # in sequences, so total rows can only shrink when the remainder is dropped
    assert sum(rows[2]) <= rows[1][0] and sum(rows[3]) <= rows[1][0]
    assert all(r > 0 for r in rows[3])


def _fd_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sbr_rs_amd.partitioned import _FdExchange

        path = os.path.join(out_dir, f"payload{rank}.bin")
        with open(path, "wb") as f:
            f.write(bytes([rank + 1]) * 1000)
        fd = os.open(path, os.O_RDONLY)
        ex = _FdExchange()
        try:
            got = ex.all_to_all({"tag": f"from{rank}"}, [fd])
        finally:
            ex.close()
        os.close(fd)
        seen = {}
        for peer, (msg, fds) in got.items():
            assert msg["tag"] == f"from{peer}" and len(fds) == 1
            seen[peer] = os.pread(fds[0], 1000, 0)
            os.close(fds[0])
        np.save(os.path.join(out_dir, f"fd{rank}.npy"), np.array([[p, data[0], len(data)] for p, data in sorted(seen.items())]))
    finally:
        dist.destroy_process_group()


def test_file_descriptor_exchange_between_ranks(tmp_path):
    """The control plane of the partitioned table / peer transport: every rank hands every other rank a file
    descriptor (SCM_RIGHTS over Unix sockets, rendezvous through torch.distributed).  Here the descriptors are
    plain files, so the plumbing is covered without a GPU."""
    world = 3
    mp.spawn(_fd_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        rows = np.load(tmp_path / f"fd{r}.npy")
        assert [int(x[0]) for x in rows] == [p for p in range(world) if p != r]
        assert all(int(x[1]) == int(x[0]) + 1 and int(x[2]) == 1000 for x in rows)
