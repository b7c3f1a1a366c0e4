"""Model persistence — the engine-side counterpart of the reference's serde derives
(`#[derive(Serialize, Deserialize)]` on Hyperparameters / Parameters / models,
/root/reference/src/models/lstm.rs:38,204,386 and ewma.rs:44,208,401; `bincode` in Cargo.toml:18).

The reference never calls a serialiser itself; what the derives promise is that a model —
hyper-parameters, parameters AND optimiser state (wyrm keeps the Adagrad/Adam accumulators inside
HogwildParameter, so `fit` continues where it stopped) — round-trips.  Here that state is the
hyper-parameter struct, every `sbr_param` block that is non-empty, and the two counters (epoch
counter keying the negative draws, optimiser step count).  Stored as a NumPy `.npz`.
"""
from __future__ import annotations

import numpy as np

from ._abi import Param, SbrHparams, make_hparams
from .engine import Model

_HP_FIELDS = ["num_items", "max_sequence_length", "embedding_dim", "learning_rate", "l2_penalty", "model", "loss",
              "optimizer", "parallelism", "num_epochs", "num_devices", "device_rank", "batch_sequences"]


def save_model(model, path: str) -> None:
    """`model` is an ImplicitLSTMModel / ImplicitEWMAModel (or a raw engine Model)."""
    eng: Model = getattr(model, "params", model)
    hp = eng.hp
    out = {f"hp_{f}": np.asarray(getattr(hp, f)) for f in _HP_FIELDS}
    out["hp_seed"] = np.frombuffer(bytes(hp.seed), dtype=np.uint8).copy()
    epoch, steps = eng.counters()
    out["counters"] = np.asarray([epoch, steps], dtype=np.uint64)
    for p in Param:
        if eng.param_count(p):
            out[f"param_{p.name}"] = eng.get_param(p)
    np.savez(path, **out)


def load_engine(path: str, device_rank: int = None) -> Model:
    z = np.load(path)
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
    return eng


def load_model(path: str):
    """Returns an ImplicitLSTMModel or ImplicitEWMAModel.  Note: the model RNG (which drives the
    shuffles of the *next* fit) restarts from the saved seed; parameters, optimiser state and
    counters are restored exactly."""
    from .ewma import ImplicitEWMAModel
    from .lstm import ImplicitLSTMModel

    eng = load_engine(path)
    return ImplicitEWMAModel(eng) if int(eng.hp.model) == 2 else ImplicitLSTMModel(eng)
