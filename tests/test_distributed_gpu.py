"""Two PROCESSES driving the HIP engine through sbr_rs_amd.distributed (the code path `torchrun`
takes for --gpus N): rendezvous, rank -> replica mapping, per-rank HipBackend, the owner-reduce
exchange and fit_distributed.  A gpurun box has one GPU and RCCL refuses two ranks on one device, so
both ranks share cuda:0 and the collectives travel over gloo with host staging
(distributed._staged_exchange); every kernel and every C-ABI call is the production one.  Each
replica must end bit-identical to the oracle emulating both devices."""
import os
import socket

import numpy as np
import pytest

from helpers import LOSS_HINGE, LOSS_WARP, hparams, synthetic_interactions
from sbr_rs_amd._abi import ModelKind, Param

PARAMS = {2: [Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.ITEM_BIAS_ACC, Param.EWMA_ALPHA],
          0: [Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.LSTM_W, Param.LSTM_W_ACC, Param.LSTM_B]}
CASE = dict(users=70, items=157, T=12, B=5, d=32, epochs=2, seed=9)


def _worker(rank, world, port, kind, loss, par, out_dir, partition=False, transport="collective"):
    os.environ["SBR_EXCHANGE_TRANSPORT"] = transport
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sbr_rs_amd as sbr
        from sbr_rs_amd._abi import ModelKind as MK

        c = CASE
        ptr, items = synthetic_interactions(c["users"], c["items"], c["T"] + 4, seed=c["seed"], zipf=True)
        comp = sbr.data.CompressedInteractions(c["users"], c["items"], ptr, items, np.zeros(len(items), dtype=np.uint64))
        if kind == int(MK.EWMA):
            h = sbr.ewma.Hyperparameters.new(c["items"], c["T"])
        else:
            h = sbr.lstm.Hyperparameters.new(c["items"], c["T"]).lstm_variant(sbr.LSTMVariant.Normal)
        model = (h.from_seed(bytes([42] * 16)).embedding_dim(c["d"]).learning_rate(0.16).l2_penalty(0.0004)
                 .loss(sbr.Loss(loss)).optimizer(sbr.Optimizer.Adagrad).num_epochs(c["epochs"]).num_threads(world)
                 .parallelism(sbr.Parallelism(par)).batch_sequences(c["B"]).partition_item_table(partition).build(device_rank=rank))
        assert model.params.is_partitioned() == partition
        loss_v = model.fit(comp)  # -> fit_distributed: a process group is initialised
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss_v,
                 **{p.name: model.params.get_param(p) for p in PARAMS[kind]})
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,loss,par", [(int(ModelKind.LSTM_NORMAL), LOSS_WARP, 1), (int(ModelKind.EWMA), LOSS_HINGE, 1),
                                           (int(ModelKind.LSTM_NORMAL), LOSS_HINGE, 0)])
def test_two_processes_share_one_gpu(tmp_path, oracle_lib, kind, loss, par):
    import torch.multiprocessing as mp

    from oracle.oracle import OracleModel

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, loss, par, str(tmp_path)), nprocs=world, join=True)
    c = CASE
    ptr, items = synthetic_interactions(c["users"], c["items"], c["T"] + 4, seed=c["seed"], zipf=True)
    ref = OracleModel(hparams(c["items"], c["T"], c["d"], kind, loss, epochs=c["epochs"], B=c["B"], ndev=world, par=par))
    ref_loss = ref.fit(ptr, items)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for p in PARAMS[kind]:
            assert np.array_equal(z[p.name].view(np.uint32), ref.get_param(p).view(np.uint32)), f"rank {r}: {p.name}"
        assert float(z["loss"]) == pytest.approx(ref_loss, rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,loss", [(int(ModelKind.EWMA), LOSS_HINGE), (int(ModelKind.LSTM_NORMAL), LOSS_WARP)])
def test_partitioned_table_two_processes(tmp_path, oracle_lib, kind, loss):
    """The item-partitioned table under one process per GPU: each process owns half of the rows, maps the
    other half from a file descriptor its peer exported (HIP virtual memory management + SCM_RIGHTS), reads
    remote rows with the unchanged kernels and updates only its own from the peers' shared gradient lists.
    Both processes sit on cuda:0 here; get_param reads the WHOLE table through each process's mapping, so
    both must return the oracle's num_devices = 2 table bit for bit."""
    import torch.multiprocessing as mp

    from oracle.oracle import OracleModel

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, loss, 1, str(tmp_path), True), nprocs=world, join=True)
    c = CASE
    ptr, items = synthetic_interactions(c["users"], c["items"], c["T"] + 4, seed=c["seed"], zipf=True)
    ref = OracleModel(hparams(c["items"], c["T"], c["d"], kind, loss, epochs=c["epochs"], B=c["B"], ndev=world))
    ref_loss = ref.fit(ptr, items)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for p in PARAMS[kind]:
            assert np.array_equal(z[p.name].view(np.uint32), ref.get_param(p).view(np.uint32)), f"rank {r}: {p.name}"
        assert float(z["loss"]) == pytest.approx(ref_loss, rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,loss", [(int(ModelKind.LSTM_NORMAL), LOSS_WARP), (int(ModelKind.EWMA), LOSS_HINGE)])
def test_peer_transport_two_processes(tmp_path, oracle_lib, kind, loss):
    """The replicated exchange with the peers' chunk buffers read in place (exported once as file
    descriptors, mapped by the peers, consumed by the owner-reduce / table-update kernels): no bulk
    collective.  Must equal the oracle's num_devices = 2 run bit for bit on both ranks."""
    import torch.multiprocessing as mp

    from oracle.oracle import OracleModel

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, loss, 1, str(tmp_path), False, "peer"), nprocs=world, join=True)
    c = CASE
    ptr, items = synthetic_interactions(c["users"], c["items"], c["T"] + 4, seed=c["seed"], zipf=True)
    ref = OracleModel(hparams(c["items"], c["T"], c["d"], kind, loss, epochs=c["epochs"], B=c["B"], ndev=world))
    ref_loss = ref.fit(ptr, items)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for p in PARAMS[kind]:
            assert np.array_equal(z[p.name].view(np.uint32), ref.get_param(p).view(np.uint32)), f"rank {r}: {p.name}"
        assert float(z["loss"]) == pytest.approx(ref_loss, rel=1e-6)


# ---- RCCL itself (backend "nccl"), world size 1 ------------------------------------------------------------
def _bench(extra):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-mrr",
           "--standalone-steps", "0", "--cold-items", "0", "--batch-sweep", "", "--param-crc", "--users", "3000", "--items", "20000",
           "--max-len", "32", "--dim", "64", "--batch-sequences", "1000"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["collective", "peer"])
def test_rccl_exchange_world_one_equals_single_device(transport):
    """The multi-GPU step through RCCL (backend nccl): `bench.py --force-exchange` runs the production step
    sequence — scatter into owner chunks, all_to_all_single / all_gather_into_tensor (collective transport) or
    the peers' buffers read in place with RCCL barriers and the dense all-gather (peer transport), owner reduce,
    table apply — on a process group of one rank.  The trained parameters must equal the plain single-device
    run's bit for bit (CRC-32 of every parameter array)."""
    plain = _bench([])
    exch = _bench(["--force-exchange", "--backend", "nccl", "--transport", transport])
    assert exch["param_crc"] == plain["param_crc"]
    assert exch["interactions_timed"] == plain["interactions_timed"]


@pytest.mark.gpu
def test_bench_gpus_2_self_launched():
    """`python bench.py --gpus 2` WITHOUT a launcher (the command the driver's SCALE run issues): bench.py starts its two ranks
    itself through torch.distributed.run; they share cuda:0 here, so the collectives travel over gloo.  One JSON line, n_gpus 2,
    a two-rank process group, and both replicas end with the same parameters (CRC-32 per array, gathered over the ranks)."""
    env_drop = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_drop if k in os.environ}
    try:
        line = _bench(["--gpus", "2", "--backend", "gloo"])
    finally:
        os.environ.update(saved)
    assert line["n_gpus"] == 2 and line["process_group_ranks"] == 2 and line["scaling"] == "weak"
    assert len(line["ms_per_step_per_rank"]) == 2
    assert line["param_crc_ranks"] == 2 and line["param_crc_ranks_equal"] is True
    assert line["steps"] == 2 and line["interactions_timed"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--partition-table", "--model", "ewma", "--loss", "hinge"]])
def test_bench_group_driver_two_replicas(extra):
    """`bench.py --driver group --gpus 2`: two replicas driven from one process through sbr_group_fit's step sequence (the path
    INTEGRATION.md binds; sequence_model.rs:90-102 inside one process), sharing cuda:0 here.  One JSON line with the driver's
    fields, host enqueue time per step for both host-thread modes, and equal parameter CRCs on both replicas."""
    env_drop = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    saved = {k: os.environ.pop(k) for k in env_drop if k in os.environ}
    try:
        line = _bench(["--driver", "group", "--gpus", "2"] + extra)
    finally:
        os.environ.update(saved)
    assert line["driver"] == "group" and line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["param_crc_replicas"] == 2 and line["param_crc_replicas_equal"] is True
    assert line["host_enqueue_ms_per_step"] > 0 and {"library_default", "one_host_thread", "host_thread_per_device"} <= set(line["modes"])
    if not extra:  # replicated table, Synchronous: the other form of the step is timed beside the default one (same models, same bits)
        assert {line["modes"]["library_default"]["exchange"], line["modes"]["library_default_other_exchange"]["exchange"]} == {"owner-applied", "gradient all-gather"}
    assert line["modes"]["host_thread_per_device"]["host_threads"] == 2 and line["modes"]["one_host_thread"]["host_threads"] == 1
    assert line["value"] > 0 and line["modes"]["library_default"]["interactions_timed"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind,loss", [(0, LOSS_WARP), (2, LOSS_HINGE)])
def test_rccl_inside_the_library_world_one(kind, loss):
    """sbr_comm_*: the step's rendezvous through RCCL opened by the library itself (no torch.distributed, no host transport) — a
    communicator of ONE rank on cuda:0 (RCCL refuses two ranks on one device, and every box of this pool has one): the whole fit
    through sbr_model_fit_comm (scatter, ncclSend / ncclRecv all-to-all, owner reduce, ncclAllGather of the reduced chunk and of the
    dense block, the two updates) must equal the plain single-device fit bit for bit, and the step-wise form
    (sbr_fit_step_local + sbr_fit_step_exchange) as well."""
    from sbr_rs_amd._abi import Param
    from sbr_rs_amd.engine import Comm, Model

    c = CASE
    ptr, items = synthetic_interactions(c["users"], c["items"], c["T"] + 4, seed=c["seed"], zipf=True)
    hp = hparams(c["items"], c["T"], c["d"], kind, loss, epochs=c["epochs"], B=c["B"])
    plain, through, stepped = Model(hp), Model(hp), Model(hp)
    comm = Comm(Comm.unique_id(), 1, 0)
    lp = plain.fit(ptr, items)
    lt = comm.fit(through, ptr, items)
    plan = stepped.fit_begin(ptr, items)
    for _ in range(c["epochs"]):
        for mb in range(plan.epoch_prepare()):
            plan.step_local(mb)
            comm.step_exchange(plan, mb)
    ls, _ = plan.end()
    assert lt == pytest.approx(lp, rel=1e-6) and ls == pytest.approx(lp, rel=1e-6)
    for p in PARAMS[kind]:
        a = plain.get_param(p)
        assert np.array_equal(a.view(np.uint32), through.get_param(p).view(np.uint32)), p.name
        assert np.array_equal(a.view(np.uint32), stepped.get_param(p).view(np.uint32)), p.name
    comm.close()
