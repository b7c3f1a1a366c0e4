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
    out["rng_state"] = np.frombuffer(eng.get_rng(), dtype=np.uint8).copy()
    for p in Param:
        if eng.param_count(p):
            out[f"param_{p.name}"] = eng.get_param(p)
    np.savez(_npz_path(path), **out)


def load_engine(path: str, device_rank: int = None) -> Model:
    z = np.load(_npz_path(path))
    kw = {f: z[f"hp_{f}"].item() for f in _HP_FIELDS}
    if device_rank is not None:
        kw["device_rank"] = device_rank
    hp: SbrHparams = make_hparams(kw["num_items"], kw["max_sequence_length"], kw["embedding_dim"], kw["learning_rate"],
                                  kw["l2_penalty"], kw["model"], kw["loss"], kw["optimizer"], kw["parallelism"],
                                  bytes(z["hp_seed"].tobytes()), kw["num_epochs"], kw["num_devices"], kw["device_rank"],
                                  kw["batch_sequences"])
    eng = Model(hp)
    for p in Param:
        key = f"param_{p.name}"
        if key in z.files:
            eng.set_param(p, z[key])
    eng.set_counters(int(z["counters"][0]), int(z["counters"][1]))
    if "rng_state" in z.files:
        eng.set_rng(z["rng_state"].tobytes())
    return eng


def load_model(path: str):
    """Returns an ImplicitLSTMModel or ImplicitEWMAModel: parameters, optimiser state, counters and the
    model RNG are restored exactly, so the next ``fit`` is the one the saved model would have run.  A model
    saved with ``num_threads(n)`` in one process comes back as its n replicas (replica r on HIP device
    r mod device count); under one process per GPU every rank loads its own replica
    (``load_engine(path, device_rank=rank)``)."""
    from .engine import device_count, set_device
    from .ewma import ImplicitEWMAModel
    from .lstm import ImplicitLSTMModel

    eng = load_engine(path, device_rank=0)
    cls = ImplicitEWMAModel if int(eng.hp.model) == 2 else ImplicitLSTMModel
    world = int(eng.hp.num_devices)
    try:
        import torch.distributed as dist

        launched = dist.is_available() and dist.is_initialized()
    except ImportError:
        launched = False
    if world == 1 or launched:
        return cls(eng)
    ndev = device_count()
    group = [eng]
    for r in range(1, world):
        set_device(r % ndev)
        group.append(load_engine(path, device_rank=r))
    set_device(0)
    return cls(eng, group=group)
