"""Model persistence — the engine-side counterpart of the reference's serde derives
(`#[derive(Serialize, Deserialize)]` on Hyperparameters / Parameters / models,
/root/reference/src/models/lstm.rs:38,204,386 and ewma.rs:44,208,401; `bincode` in Cargo.toml:18).

The reference never calls a serialiser itself; what the derives promise is that a model —
hyper-parameters, parameters AND optimiser state (wyrm keeps the Adagrad/Adam accumulators inside
HogwildParameter, so `fit` continues where it stopped) — round-trips.  Here that state is the
hyper-parameter struct (with the state of its RNG, which the reference's Hyperparameters serialise:
the next `fit` continues the shuffle / partition-seed stream), every `sbr_param` block that is
non-empty, and the two counters (epoch counter keying the negative draws, optimiser step count).
Stored as a NumPy `.npz`.  A model built with `num_threads(n)` (n replicas in one process) is
restored as n replicas, every one set to the saved state — replicas are bit-identical by
construction, so one copy of the parameters is enough.
"""
from __future__ import annotations

import numpy as np

from ._abi import Param, SbrHparams, make_hparams
from .engine import Model

_HP_FIELDS = ["num_items", "max_sequence_length", "embedding_dim", "learning_rate", "l2_penalty", "model", "loss",
              "optimizer", "parallelism", "num_epochs", "num_devices", "device_rank", "batch_sequences"]


def _npz_path(path: str) -> str:
    return path if str(path).endswith(".npz") else str(path) + ".npz"  # np.savez appends the suffix; keep load symmetric


def save_model(model, path: str) -> None:
    """`model` is an ImplicitLSTMModel / ImplicitEWMAModel (or a raw engine Model)."""
    eng: Model = getattr(model, "params", model)
    hp = eng.hp
    out = {f"hp_{f}": np.asarray(getattr(hp, f)) for f in _HP_FIELDS}
    out["hp_seed"] = np.frombuffer(bytes(hp.seed), dtype=np.uint8).copy()
    epoch, steps = eng.counters()
    out["counters"] = np.asarray([epoch, steps], dtype=np.uint64)
    out["partitioned"] = np.asarray(1 if eng.is_partitioned() else 0)  # the item table stored once across the replicas
    out["rng_state"] = np.frombuffer(eng.get_rng(), dtype=np.uint8).copy()
    for p in Param:
        if eng.param_count(p):
            out[f"param_{p.name}"] = eng.get_param(p)
    np.savez(_npz_path(path), **out)


_TABLE_PARAMS = ("ITEM_EMBEDDING", "ITEM_BIAS", "ITEM_EMBEDDING_ACC", "ITEM_BIAS_ACC", "ITEM_EMBEDDING_M", "ITEM_BIAS_M")


def _hparams_of(z, device_rank=None) -> SbrHparams:
    kw = {f: z[f"hp_{f}"].item() for f in _HP_FIELDS}
    if device_rank is not None:
        kw["device_rank"] = device_rank
    return make_hparams(kw["num_items"], kw["max_sequence_length"], kw["embedding_dim"], kw["learning_rate"],
                        kw["l2_penalty"], kw["model"], kw["loss"], kw["optimizer"], kw["parallelism"],
                        bytes(z["hp_seed"].tobytes()), kw["num_epochs"], kw["num_devices"], kw["device_rank"],
                        kw["batch_sequences"])


def _restore_into(eng: Model, z, table: bool = True) -> Model:
    """Parameters, optimiser state, counters and RNG of a saved model into `eng`; table = False skips the item-table
    blocks (a partitioned group stores them once: one replica writes them)."""
    for p in Param:
        key = f"param_{p.name}"
        if key in z.files and (table or p.name not in _TABLE_PARAMS):
            eng.set_param(p, z[key])
    eng.set_counters(int(z["counters"][0]), int(z["counters"][1]))
    if "rng_state" in z.files:
        eng.set_rng(z["rng_state"].tobytes())
    return eng


def load_engine(path: str, device_rank: int = None) -> Model:
    z = np.load(_npz_path(path))
    return _restore_into(Model(_hparams_of(z, device_rank)), z)


def load_model(path: str):
    """Returns an ImplicitLSTMModel or ImplicitEWMAModel: parameters, optimiser state, counters and the
    model RNG are restored exactly, so the next ``fit`` is the one the saved model would have run.  A model
    saved with ``num_threads(n)`` in one process comes back as its n replicas (replica r on HIP device
    r mod device count); under one process per GPU (torch.distributed initialised, world = n) every rank gets ITS
    replica (device_rank = its rank).  A model whose item table was partitioned is rebuilt partitioned — one copy of
    the table across the replicas / ranks, never n full tables."""
    from .engine import device_count, set_device
    from .ewma import ImplicitEWMAModel
    from .lstm import ImplicitLSTMModel

    try:
        import torch.distributed as dist

        launched = dist.is_available() and dist.is_initialized()
    except ImportError:
        launched = False
    z = np.load(_npz_path(path))
    world = int(z["hp_num_devices"].item())
    partitioned = "partitioned" in z.files and int(z["partitioned"].item()) == 1
    if launched and world > 1 and dist.get_world_size() == world:
        # (a process group of another size — e.g. one left over from something else — loads the replicas in this process, below)
        rank = dist.get_rank()  # one process per GPU: every rank restores ITS replica
        if partitioned:
            from .partitioned import create_partitioned_model

            # the table exists once and is mapped by every rank: rank 0 alone writes its blocks, and nobody returns (and
            # starts reading rows) before that write has finished
            eng = _restore_into(create_partitioned_model(_hparams_of(z, rank)), z, table=rank == 0)
            eng.synchronize()
            dist.barrier()
        else:
            eng = load_engine(path, device_rank=rank)
        return (ImplicitEWMAModel if int(eng.hp.model) == 2 else ImplicitLSTMModel)(eng)
    if partitioned and world > 1:
        # one process, n replicas over ONE copy of the table: rebuild the group, not n full tables
        from .engine import group_create

        group = group_create(_hparams_of(z, 0), world, partition_item_table=True)
        for r, g in enumerate(group):
            _restore_into(g, z, table=r == 0)  # the table arrays are shared: written once
        cls = ImplicitEWMAModel if int(group[0].hp.model) == 2 else ImplicitLSTMModel
        return cls(group[0], group=group)
    eng = load_engine(path, device_rank=0)
    cls = ImplicitEWMAModel if int(eng.hp.model) == 2 else ImplicitLSTMModel
    if world == 1:
        return cls(eng)
    ndev = device_count()
    group = [eng]
    for r in range(1, world):
        set_device(r % ndev)
        group.append(load_engine(path, device_rank=r))
    set_device(0)
    return cls(eng, group=group)
