#!/usr/bin/env python
"""Test MRR against the minibatch size on a synthetic catalogue with PLANTED sequential structure, large enough
for the batch sizes the throughput bench uses (MovieLens-100K has 1.3 K training subsequences, so its sweep in
tools/movielens_batch_sweep.py stops at batch 1 024).

Data: `items` items in 500 clusters; a session stays in its cluster with probability `p_follow` per step (next item
Zipf(1) over the cluster's members) and otherwise jumps to a random cluster — so a model that learns which items
belong together, and which of them are popular, ranks the held-out last item near the top, and one that learns
nothing ranks it at ~items/2 (MRR ~ 2e-4).  `users` training users, 2 000 held-out
users (the reference's protocol: unseen users, history = all but the last item, evaluation.rs:12-48).

    tools/planted_batch_sweep.py [--users 200000] [--items 50000] [--batches 16,256,4096,16384,50000] [--seeds 3]
                                 [--schedules 50000:0.32:5,50000:0.16:15]   (batch:learning rate:epochs, extra rows)
                                 [--json profiles/quality_neutral_batch.json]

Runs on the GPU engine (this is a statement about the optimisation regime, not a parity test).  Prints one
markdown table (mean +- sd over the model seeds); the numbers are quoted in NOTES.md section 3.  --json writes the
largest batch whose mean LSTM test MRR is within 3 % of the smallest batch's — what bench.py reports as
`value_quality_neutral`.
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
    ap.add_argument("--batches", type=str, default="256,1024,4096,8192,16384,32768,50000")
    ap.add_argument("--models", type=str, default="lstm,ewma")
    ap.add_argument("--seeds", type=int, default=3, help="model seeds per row (initialisation, shuffles, negative draws)")
    ap.add_argument("--schedules", type=str, default="", help="extra LSTM rows batch:lr:epochs[,...] (large-batch schedules)")
    ap.add_argument("--json", type=str, default="", help="write the quality-neutral batch (LSTM, within 3 %% of the smallest batch) here")
    a = ap.parse_args()
    import json

    from sbr_rs_amd._abi import make_hparams
    from sbr_rs_amd.engine import Model

    ptr, it = planted(a.users, a.items, a.max_len, a.p_follow, 1)
    tptr, tit = planted(2000, a.items, a.max_len, a.p_follow, 2)
    print(f"planted catalogue: {a.users} users, {a.items} items, len 8..{a.max_len}, p_follow {a.p_follow}, "
          f"{int(ptr[-1]) - a.users} training interactions, dim {a.dim}, WARP, Adagrad l2 4e-4, {a.seeds} model seeds per row, "
          f"2000 held-out users")
    print("| model | batch_sequences | learning rate | epochs | optimiser steps / epoch | test MRR mean +- sd (min .. max) | fit s |")
    print("|---|---|---|---|---|---|---|")

    def row(name, b, lr, epochs):
        kind = {"lstm": 0, "ewma": 2}[name]
        mrrs, dts = [], []
        for sd in range(a.seeds):
            hp = make_hparams(a.items, a.max_len, a.dim, lr, 0.0004, kind, 2, 0, 1, bytes([42 + sd] * 16), epochs, 1, 0, b)
            m = Model(hp)
            t0 = time.perf_counter()
            m.fit(ptr, it)
            dts.append(time.perf_counter() - t0)
            mrrs.append(float(m.mrr_score(tptr, tit)[0]))
            m.close() if hasattr(m, "close") else None
        mean, sdv = float(np.mean(mrrs)), float(np.std(mrrs, ddof=1)) if len(mrrs) > 1 else 0.0
        print(f"| {name} | {b} | {lr} | {epochs} | {-(-a.users // b)} | {mean:.4f} +- {sdv:.4f} ({min(mrrs):.4f} .. {max(mrrs):.4f}) | {np.mean(dts):.1f} |", flush=True)
        return mean, sdv

    results = {}
    for name in a.models.split(","):
        for b in [int(x) for x in a.batches.split(",")]:
            results[name, b] = row(name, b, 0.16, a.epochs)
    for spec in [x for x in a.schedules.split(",") if x]:
        b, lr, ep = spec.split(":")
        row("lstm", int(b), float(lr), int(ep))
    if a.json and any(k[0] == "lstm" for k in results):
        bs = sorted(b for (n, b) in results if n == "lstm")
        base = results["lstm", bs[0]][0]
        ok = [b for b in bs if results["lstm", b][0] >= 0.97 * base]
        neutral = max(b for b in ok if all(results["lstm", x][0] >= 0.97 * base for x in bs if x <= b))
        json.dump({"batch_sequences": neutral, "criterion": f"largest batch whose mean LSTM test MRR over {a.seeds} seeds stays within 3 % of "
                   f"batch {bs[0]}'s on the planted-structure catalogue (and every smaller batch does too)",
                   "reference_batch": bs[0], "reference_mrr": base,
                   "rows": [{"batch_sequences": b, "mrr_mean": results["lstm", b][0], "mrr_sd": results["lstm", b][1]} for b in bs],
                   "table": "profiles/r03_planted_batch_sweep.md", "tool": "tools/planted_batch_sweep.py"}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
