#!/usr/bin/env python
"""Test MRR of the reference's five MovieLens-100K cases (lstm.rs:450-520, ewma.rs:463-507) on the CPU oracle
(batch_sequences = 1 = the reference's per-sequence SGD), as a DISTRIBUTION: the reference's protocol fixes one
seed, and test MRR over the 188 test users moves by about +-0.01 with anything that changes the order of the
updates — which is why the reference itself carries two thresholds per case (default / MKL_CBWR=AVX).

    tools/mrr_stream_sweep.py [--streams 24] [--split fixed|varying] [--jobs 8] [--reference-order]

--reference-order  the oracle's checker-only mode: negatives from the worker's sequential xorshift stream and, for the
                 2-thread case, one Adagrad application per worker — the reference's order of work instead of the
                 contract's counter-keyed draws / summed-gradient update (oracle/sbr_oracle.c, `reference_order`).

--split fixed    the reference's split (XorShiftRng::from_seed([42; 16]) -> user_based_split 0.2); stream k > 0
                 advances the already-advanced RNG by 7k draws before it is moved into the model, so only the
                 model's streams (initialisation, shuffles, partition seeds) change.  Stream 0 IS the protocol.
--split varying  seed [k; 16] for the split as well (different test users every time).

Prints mean / sd / min / max per case next to the reference's bounds.  Results of this script are quoted in
NOTES.md section 3.
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"lstm hinge 1 thread": (0, 1, 1, (0.081, 0.091)), "lstm hinge 2 threads": (0, 1, 2, (0.074, 0.078)),
         "lstm warp": (0, 2, 1, (0.10, 0.089)), "ewma hinge": (2, 1, 1, (0.11, 0.091)), "ewma warp": (2, 2, 1, (0.14, 0.089))}


def one(case, k, split, reference_order=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import hparams, load_movielens
    from oracle.oracle import OracleModel
    from sbr_rs_amd.data import user_based_split
    from sbr_rs_amd.rng import XorShiftRng

    kind, loss, threads, _ = CASES[case]
    data = load_movielens()
    rng = XorShiftRng.from_seed(bytes([42 if (split == "fixed" or k == 0) else k] * 16))
    train, test = user_based_split(data, rng, 0.2)
    if split == "fixed":
        for _ in range(7 * k):
            rng.next_u32()
    train, test = train.to_compressed(), test.to_compressed()
    m = OracleModel(hparams(data.num_items(), 128, 32, kind, loss, epochs=10, B=1, seed=rng.state_seed(), ndev=threads))
    if reference_order:
        m.set_reference_order(True)
    m.fit(train.user_pointers, train.item_ids)
    return float(m.mrr_score(test.user_pointers, test.item_ids)[0])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        print(one(sys.argv[2], int(sys.argv[3]), sys.argv[4], len(sys.argv) > 5 and sys.argv[5] == "ref"))
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=24)
    ap.add_argument("--split", choices=["fixed", "varying"], default="fixed")
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--reference-order", action="store_true")
    a = ap.parse_args()
    jobs = [(c, k) for c in CASES for k in range(a.streams)]
    res, running = {c: {} for c in CASES}, []

    def drain():
        for it in list(running):
            if it[2].poll() is not None:
                res[it[0]][it[1]] = float(it[2].communicate()[0].strip().splitlines()[-1])
                running.remove(it)

    for c, k in jobs:
        while len(running) >= a.jobs:
            drain()
            time.sleep(0.05)
        running.append((c, k, subprocess.Popen([sys.executable, __file__, "--one", c, str(k), a.split] + (["ref"] if a.reference_order else []),
                                               stdout=subprocess.PIPE, text=True)))
    while running:
        drain()
        time.sleep(0.05)
    print(f"mode: {'reference order (sequential negative stream; one update per worker)' if a.reference_order else 'contract (counter-keyed negatives; one update from the summed gradients)'}, split {a.split}")
    print(f"| case | reference bound (default / CI) | protocol run (stream 0) | mean of {a.streams} | sd | min | max |")
    print("|---|---|---|---|---|---|---|")
    for c, (_, _, _, b) in CASES.items():
        v = np.array([res[c][k] for k in range(a.streams)])
        print(f"| {c} | {b[0]} / {b[1]} | {v[0]:.4f} | {v.mean():.4f} | {v.std(ddof=1):.4f} | {v.min():.4f} | {v.max():.4f} |")
