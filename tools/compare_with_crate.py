#!/usr/bin/env python
"""Compares the sbr crate's dump (integration/rust_check: `cargo run --release -- data.csv crate_dump.json`) with this
repository's engine-side dumps of the same five MovieLens protocol cases in reference order (tests/golden/reference_order_*.npz;
tools/make_reference_order_golden.py) and reports, stream by stream, whether they agree and where they first part.

    python tools/compare_with_crate.py crate_dump.json

The streams are ordered from the bottom up — generator, split, model-RNG replay, shuffle, worker seeds, visiting order, raw
negative draws, chosen negatives, test ranks, MRR, returned loss — so the FIRST line that says DIFFERS names the lowest layer
whose recollection (wyrm 0.9.1 and rand 0.5 were not available when the engine was written) is wrong.
Exit status 0: index streams and ranks all agree (the pin DESIGN.md section 3 lacks); 1: something differs.  Floats (MRR, loss)
are reported with their difference; they are expected to agree only to float tolerance even when every index agrees, because
wyrm's arithmetic (fast-math, its own reduction orders) is not the engine's contract.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def first_difference(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    n = min(a.size, b.size)
    bad = np.flatnonzero(a[:n].astype(np.int64) != b[:n].astype(np.int64)) if n else np.array([], dtype=np.int64)
    if bad.size:
        return int(bad[0])
    return None if a.size == b.size else n


def report(label, crate, ours, out):
    d = first_difference(crate, ours)
    same = d is None
    out.append(same)
    if same:
        print(f"  agrees   {label} ({np.asarray(ours).size} values)")
    else:
        c, o = np.asarray(crate).ravel(), np.asarray(ours).ravel()
        print(f"  DIFFERS  {label}: first at element {d}: crate {c[d] if d < c.size else '<end>'} / engine {o[d] if d < o.size else '<end>'}"
              f" (lengths {c.size} / {o.size})")


def main(path):
    dump = json.load(open(path))
    verdict = []
    print(f"crate: {dump.get('crate_name')}")
    s, g = dump["streams"], np.load(os.path.join(GOLDEN, "reference_order_streams.npz"))
    print("generator streams (rand 0.5, SURVEY App. C):")
    for k in ("next_u32", "uniform_u64", "uniform_usize_1683", "shuffle_10", "normal_bits", "gen_seed16"):
        report(k, np.array(s[k], dtype=np.uint64), g[k].astype(np.uint64), verdict)
    print("user_based_split (data.rs:69-88) on data.csv:")
    sp = dump["split"]
    for k in ("train_users_with_data", "test_users_with_data", "train_interactions", "test_interactions", "train_items_fnv", "test_items_fnv"):
        report(k, np.array([sp[k]], dtype=np.uint64), np.array([g[k]], dtype=np.uint64), verdict)
    for case in dump["cases"]:
        name = case["name"]
        ours = np.load(os.path.join(GOLDEN, "reference_order_" + name.replace(" ", "_") + ".npz"))
        rp = case["replay"]
        print(f"{name}" + (f"  [replay assumes {rp['assumed_wyrm_lstm_draws']} normal draws inside wyrm's lstm::Parameters::new]" if rp["assumed_wyrm_lstm_draws"] else ""))
        report("number of subsequences", [rp["num_subsequences"]], [int(ours["num_subsequences"])], verdict)
        report("order after the model generator's shuffle (sequence_model.rs:84)", np.array(rp["shuffled_order"]), ours["shuffled_order"], verdict)
        report("worker seeds (:97)", np.array(rp["worker_seeds"]), ours["worker_seeds"], verdict)
        report("first epoch's visiting order (:109)", np.array(rp["first_epoch_order"]), ours["first_epoch_order"], verdict)
        report("workers' raw negative draws (:58-65, :137)", np.array(rp["first_epoch_raw_draws"]), ours["first_epoch_raw_draws"], verdict)
        # the engine's chosen negatives against the CRATE's raw draws: term t takes tries[t] draws, the last one is kept
        tries, negs = ours["tries"].astype(np.int64), ours["negatives"]
        pos = np.cumsum(tries) - 1
        raw = np.array(rp["first_epoch_raw_draws"][0])
        ok = pos < raw.size
        report("engine's chosen negatives = the crate's draws at the engine's trip counts (worker 0)", raw[pos[ok]], negs[ok], verdict)
        report("test ranks (evaluation.rs:20-43)", np.array(case["test_ranks"]), ours["test_ranks"], verdict)
        for label, c, o in (("test MRR", case["test_mrr"], float(ours["test_mrr"])), ("loss returned by fit (sequence_model.rs:157, 173-177)", case["fit_loss"], float(ours["fit_loss_lagged"]))):
            print(f"  float    {label}: crate {c:.7f} / engine {o:.7f}  (difference {c - o:+.2e})")
    ok = all(verdict)
    print("RESULT: every index stream and every rank agrees — the oracle is pinned to the crate at index level" if ok else
          "RESULT: the first DIFFERS line above is the lowest layer that is not the crate's")
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    raise SystemExit(main(sys.argv[1]))
