#!/usr/bin/env python
"""The reference's own Criterion bench (benches/benchmark.rs:16-71) against the engine: `fit` on 10 000 MovieLens-100K interactions
sampled without replacement, max_sequence_length 128, embedding_dim 32, hinge, Adagrad, lr 0.16, l2 0.0004, 3 epochs, one worker —
for the LSTM and for EWMA, ten samples each like `Criterion::default().sample_size(10)`, every sample one more `fit` call on the
same model (the bench's closure re-fits the model it built once, `:40-42`).  The reference publishes no number for it (BASELINE.md
section 2); here it is timed at the reference's schedule (one optimiser step per subsequence, `batch_sequences` 1) and at 16
sequences per step, with the C oracle's time for the same fit on one host core beside it (`--oracle`).

    python tools/criterion_bench.py [--oracle] [--samples 10] [--batches 1,16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sbr_rs_amd as sbr  # noqa: E402


def sample_data(sample_size: int):
    """load_movielens (benches/benchmark.rs:16-24): a uniform sample without replacement of the CSV's rows (seeded here)."""
    data = sbr.datasets.download_movielens_100k()
    u, i, t = data.arrays()
    pick = np.sort(np.random.RandomState(20180823).choice(u.shape[0], size=sample_size, replace=False))
    return sbr.data.Interactions.from_arrays(u[pick], i[pick], t[pick]).to_compressed()


def build(kind: str, num_items: int, batch: int):
    hp = (sbr.lstm if kind == "lstm" else sbr.ewma).Hyperparameters.new(num_items, 128)
    hp = (hp.embedding_dim(32).learning_rate(0.16).l2_penalty(0.0004).loss(sbr.Loss.Hinge).optimizer(sbr.Optimizer.Adagrad)
          .num_epochs(3).num_threads(1).from_seed(bytes([42] * 16)))
    if kind == "lstm":
        hp = hp.lstm_variant(sbr.LSTMVariant.Normal)  # benchmark.rs leaves the default (Normal, lstm.rs:63)
    return hp.batch_sequences(batch).build()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--batches", type=str, default="1,16")
    ap.add_argument("--oracle", action="store_true", help="also time the C oracle (test infrastructure) on one host core")
    a = ap.parse_args()
    data = sample_data(10_000)
    out = []
    for kind in ("lstm", "ewma"):
        for batch in (int(b) for b in a.batches.split(",")):
            model = build(kind, data.num_items(), batch)
            model.fit(data)  # Criterion's warm-up
            times = []
            for _ in range(a.samples):
                t0 = time.perf_counter()
                loss = model.fit(data)
                times.append(time.perf_counter() - t0)
            row = {"bench": kind, "batch_sequences": batch, "samples": a.samples, "fit_ms_mean": 1e3 * float(np.mean(times)),
                   "fit_ms_min": 1e3 * min(times), "fit_ms_max": 1e3 * max(times), "last_loss": loss,
                   "interactions": 10_000, "epochs_per_fit": 3}
            if a.oracle and batch == 1:
                from oracle.oracle import OracleModel

                o = OracleModel(model.params.hp)
                o.fit(data.user_pointers, data.item_ids)
                t0 = time.perf_counter()
                o.fit(data.user_pointers, data.item_ids)
                row["oracle_one_core_fit_ms"] = 1e3 * (time.perf_counter() - t0)
            out.append(row)
            print(json.dumps(row), flush=True)
    return out


if __name__ == "__main__":
    main()
