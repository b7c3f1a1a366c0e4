"""Shared test helpers: fixtures and the reference's evaluation protocol."""
from __future__ import annotations

import os

import numpy as np

from sbr_rs_amd._abi import ModelKind, make_hparams
from sbr_rs_amd.data import Interactions, user_based_split
from sbr_rs_amd.rng import XorShiftRng

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

LOSS_BPR, LOSS_HINGE, LOSS_WARP = 0, 1, 2
OPT_ADAGRAD, OPT_ADAM = 0, 1
PAR_ASYNC, PAR_SYNC = 0, 1


def load_movielens() -> Interactions:
    """The reference's test data (download_movielens_100k, datasets.rs:66-71) from the fixture."""
    z = np.load(os.path.join(GOLDEN, "movielens_100k.npz"))
    return Interactions.from_arrays(z["user_id"], z["item_id"], z["timestamp"])


def movielens_protocol():
    """run_test's set-up (lstm.rs:427-434): seed [42;16], user_based_split(0.2), and the SAME,
    already advanced RNG is then moved into the model (hyperparameters.rng(rng))."""
    data = load_movielens()
    rng = XorShiftRng.from_seed(bytes([42] * 16))
    train, test = user_based_split(data, rng, 0.2)
    return data, train.to_compressed(), test.to_compressed(), rng


def synthetic_interactions(num_users, num_items, max_len, seed=7, min_len=3, zipf=False):
    """Synthetic CSR in the shape BASELINE.md §3 describes: len_u ~ U{min_len..max_len}, items
    uniform (or Zipf(1.0) over a random permutation), timestamps = position."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(min_len, max_len + 1, size=num_users).astype(np.uint64)
    ptr = np.zeros(num_users + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum(lens)
    nnz = int(ptr[-1])
    if zipf:
        ranks = np.arange(1, num_items + 1, dtype=np.float64)
        p = 1.0 / ranks
        p /= p.sum()
        perm = rs.permutation(num_items)
        items = perm[rs.choice(num_items, size=nnz, p=p)].astype(np.uint32)
    else:
        items = rs.randint(0, num_items, size=nnz).astype(np.uint32)
    return ptr, items


def hparams(num_items, T, dim, model, loss, lr=0.16, l2=0.0004, epochs=1, B=8, seed=bytes([42] * 16), ndev=1, rank=0,
            opt=OPT_ADAGRAD, par=PAR_SYNC):
    return make_hparams(num_items, T, dim, lr, l2, model, loss, opt, par, seed, epochs, ndev, rank, B)
