#!/usr/bin/env python
"""MovieLens-100K under the reference protocol (lstm.rs:427-448) for several values of the engine's
minibatch (batch_sequences; 1 = the reference's per-sequence SGD): test MRR, fit time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from helpers import movielens_protocol  # noqa: E402
from sbr_rs_amd._abi import make_hparams  # noqa: E402
from sbr_rs_amd.engine import Model  # noqa: E402

data, train, test, rng = movielens_protocol()
print("| model | loss | batch_sequences | test MRR | fit s |\n|---|---|---|---|---|")
for kind, kname in ((0, "LSTM"), (2, "EWMA")):
    for loss, lname in ((1, "hinge"), (2, "WARP")):
        for B in (1, 16, 128, 1024):
            hp = make_hparams(data.num_items(), 128, 32, 0.16, 0.0004, kind, loss, 0, 1, rng.state_seed(), 10, 1, 0, B)
            m = Model(hp)
            t0 = time.perf_counter()
            m.fit(train.user_pointers, train.item_ids)
            dt = time.perf_counter() - t0
            mrr, _ = m.mrr_score(test.user_pointers, test.item_ids)
            print(f"| {kname} | {lname} | {B} | {mrr:.4f} | {dt:.2f} |", flush=True)
            m.close()
