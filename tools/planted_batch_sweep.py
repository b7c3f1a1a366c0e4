#!/usr/bin/env python
"""Test MRR against the minibatch size on a synthetic catalogue with PLANTED sequential structure, large enough
for the batch sizes the throughput bench uses (MovieLens-100K has 1.3 K training subsequences, so its sweep in
tools/movielens_batch_sweep.py stops at batch 1 024).

Data: `items` items in 500 clusters; a session stays in its cluster with probability `p_follow` per step (next item
Zipf(1) over the cluster's members) and otherwise jumps to a random cluster — so a model that learns which items
belong together, and which of them are popular, ranks the held-out last item near the top, and one that learns
nothing ranks it at ~items/2 (MRR ~ 2e-4).  `users` training users, 2 000 held-out
users (the reference's protocol: unseen users, history = all but the last item, evaluation.rs:12-48).

    tools/planted_batch_sweep.py [--users 200000] [--items 50000] [--batches 16,256,4096,16384,50000]

Runs on the GPU engine (this is a statement about the optimisation regime, not a parity test).  Prints one
markdown table; the numbers are quoted in DESIGN.md section 3.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def planted(users, items, max_len, p_stay, seed, clusters=500):
    """Items live in `clusters` equal clusters; a session walks inside a cluster (next item Zipf(1) over the
    cluster's members) and jumps to a random cluster with probability 1 - p_stay per step."""
    rs = np.random.RandomState(seed)
    size = items // clusters
    member = np.random.RandomState(12345).permutation(items)[: clusters * size].reshape(clusters, size)  # same catalogue for train / test
    cdf = np.cumsum(1.0 / np.arange(1, size + 1))
    cdf /= cdf[-1]
    lens = rs.randint(8, max_len + 1, size=users)
    ptr = np.zeros(users + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum(lens)
    out = np.empty(int(ptr[-1]), dtype=np.uint32)
    cl = rs.randint(0, clusters, size=users)
    pos = ptr[:-1].astype(np.int64).copy()
    alive = np.arange(users)
    for t in range(max_len):
        alive = alive[lens[alive] > t]
        if alive.size == 0:
            break
        jump = rs.random_sample(alive.size) >= p_stay
        cl[alive] = np.where(jump, rs.randint(0, clusters, size=alive.size), cl[alive])
        rank = np.searchsorted(cdf, rs.random_sample(alive.size))
        out[pos[alive] + t] = member[cl[alive], np.minimum(rank, size - 1)]
    return ptr, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--p-follow", type=float, default=0.9)
    ap.add_argument("--batches", type=str, default="16,256,4096,16384,50000")
    ap.add_argument("--models", type=str, default="lstm,ewma")
    a = ap.parse_args()
    from sbr_rs_amd._abi import make_hparams
    from sbr_rs_amd.engine import Model

    ptr, it = planted(a.users, a.items, a.max_len, a.p_follow, 1)
    tptr, tit = planted(2000, a.items, a.max_len, a.p_follow, 2)
    print(f"planted catalogue: {a.users} users, {a.items} items, len 8..{a.max_len}, p_follow {a.p_follow}, "
          f"{int(ptr[-1]) - a.users} training interactions, dim {a.dim}, {a.epochs} epochs, WARP, Adagrad lr 0.16 l2 4e-4")
    print("| model | batch_sequences | optimiser steps / epoch | test MRR | fit s |")
    print("|---|---|---|---|---|")
    for name in a.models.split(","):
        kind = {"lstm": 0, "ewma": 2}[name]
        for b in [int(x) for x in a.batches.split(",")]:
            hp = make_hparams(a.items, a.max_len, a.dim, 0.16, 0.0004, kind, 2, 0, 1, bytes([42] * 16), a.epochs, 1, 0, b)
            m = Model(hp)
            t0 = time.perf_counter()
            m.fit(ptr, it)
            dt = time.perf_counter() - t0
            mrr, _ = m.mrr_score(tptr, tit)
            print(f"| {name} | {b} | {-(-a.users // b)} | {mrr:.4f} | {dt:.1f} |", flush=True)
            m.close() if hasattr(m, "close") else None


if __name__ == "__main__":
    main()
