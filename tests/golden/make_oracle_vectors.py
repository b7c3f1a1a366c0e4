"""Generates tests/golden/oracle_vectors.npz: inputs and expected outputs of the hot path for a few
small cases, produced by the CPU oracle (oracle/sbr_oracle.c).

No reference test pins any float value, RNG output, gradient or rank, and the Rust
reference cannot be built or imported here, so these vectors are the oracle's own ("parity unpinned"
below the MRR-threshold level — see the header of oracle/sbr_oracle.c).  They exist to (a) detect any
drift of the oracle itself (tests/test_oracle.py::test_oracle_reproduces_committed_vectors) and (b) let
the GPU suite check the HIP engine against committed numbers without executing the oracle
(tests/test_parity_gpu.py::test_engine_reproduces_committed_vectors).

Run: python tests/golden/make_oracle_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

from helpers import LOSS_BPR, LOSS_HINGE, LOSS_WARP, OPT_ADAM, PAR_ASYNC, PAR_SYNC, hparams, synthetic_interactions  # noqa: E402
from oracle.oracle import OracleModel  # noqa: E402
from sbr_rs_amd._abi import ModelKind, Param  # noqa: E402

# name: (kind, loss, dim, items, users, T, B, epochs, ndev, opt, par)
CASES = {
    "ewma_hinge_d32": (ModelKind.EWMA, LOSS_HINGE, 32, 61, 40, 10, 6, 2, 1, 0, PAR_SYNC),
    "lstm_warp_d32": (ModelKind.LSTM_NORMAL, LOSS_WARP, 32, 83, 50, 12, 7, 2, 1, 0, PAR_SYNC),
    "coupled_bpr_adam_d16": (ModelKind.LSTM_COUPLED, LOSS_BPR, 16, 47, 40, 9, 4, 3, 1, OPT_ADAM, PAR_SYNC),
    "lstm_hinge_two_devices": (ModelKind.LSTM_NORMAL, LOSS_HINGE, 16, 71, 60, 10, 5, 2, 2, 0, PAR_SYNC),
    "ewma_warp_three_devices_async": (ModelKind.EWMA, LOSS_WARP, 64, 97, 70, 11, 4, 2, 3, 0, PAR_ASYNC),
}
BLOCKS = [Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.ITEM_BIAS_ACC, Param.LSTM_W, Param.LSTM_B,
          Param.EWMA_ALPHA]


def case_inputs(name):
    kind, loss, d, items, users, T, B, epochs, ndev, opt, par = CASES[name]
    ptr, it = synthetic_interactions(users, items, T + 4, seed=sum(map(ord, name)) % 1000, zipf=True)
    tptr, tit = synthetic_interactions(20, items, T + 2, seed=7 + len(name))
    hp = hparams(items, T, d, int(kind), loss, epochs=epochs, B=B, ndev=ndev, opt=opt, par=par,
                 lr=0.02 if opt == OPT_ADAM else 0.16)
    return hp, (ptr, it), (tptr, tit)


def run_case(model_factory, name):
    hp, (ptr, it), (tptr, tit) = case_inputs(name)
    m = model_factory(hp)
    out = {"loss": np.float32(m.fit(ptr, it))}
    for p in BLOCKS:
        if m.param_count(p):
            out[p.name] = m.get_param(p)
    mrr, ranks = m.mrr_score(tptr, tit)
    out["mrr"] = np.float32(mrr)
    out["ranks"] = np.asarray(ranks, dtype=np.uint32)
    out["user_representation"] = m.user_representation(tit[: int(tptr[1])])
    return out


if __name__ == "__main__":
    blob = {}
    for name in CASES:
        for k, v in run_case(OracleModel, name).items():
            blob[f"{name}/{k}"] = v
    path = os.path.join(HERE, "oracle_vectors.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes,", len(blob), "arrays")
