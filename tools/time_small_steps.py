#!/usr/bin/env python
"""Fit times of small optimiser steps — the numbers of profiles/r03_small_steps.md.

    tools/time_small_steps.py movielens [epochs] [max_len]      MovieLens-100K, LSTM Normal, d = 32, WARP, Adagrad, batch_sequences 1
                                                                (BASELINE.json configs[1] at the reference's own schedule)
    tools/time_small_steps.py synthetic [dim] [B1,B2,...]       synthetic 100 000 items, len <= 64: ms per step of model.fit

SBR_WAVE=0 / SBR_DW_BLOCK=0 force the MFMA tile kernels (sbr_wave.hip's forms are bit-identical to them).  Under
`rocprofv3 --kernel-trace` the step timelines come from tools/rocpd_timeline.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def movielens(epochs=10, max_len=128):
    from helpers import movielens_protocol
    from sbr_rs_amd._abi import make_hparams
    from sbr_rs_amd.engine import Model

    data, train, _test, rng = movielens_protocol()
    hp = make_hparams(data.num_items(), max_len, 32, 0.16, 0.0004, 0, 2, 0, 1, rng.state_seed(), epochs, 1, 0, 1)
    m = Model(hp)
    t0 = time.perf_counter()
    loss = m.fit(train.user_pointers, train.item_ids)
    print(f"MovieLens-100K, {epochs} epochs, max_sequence_length {max_len}, batch_sequences 1: fit {time.perf_counter() - t0:.3f} s, loss {loss!r}")


def synthetic(dim=32, batches=(16, 64, 256, 1024, 4096)):
    from bench import synthetic_csr
    from sbr_rs_amd._abi import make_hparams
    from sbr_rs_amd.engine import Model

    users, items, T = 20000, 100000, 64
    ptr, it = synthetic_csr(users, items, T)
    for B in batches:
        u = min(users, max(B * 40, 2000))
        p, i = ptr[: u + 1], it[: int(ptr[u])]
        m = Model(make_hparams(items, T, dim, 0.16, 0.0004, 0, 2, 0, 1, bytes([7] * 16), 1, 1, 0, B))
        m.fit(p, i)  # warm-up epoch (allocations)
        t0 = time.perf_counter()
        m.fit(p, i)
        dt = time.perf_counter() - t0
        steps = -(-u // B)
        print(f"d {dim}, {B} sequences per step: {dt / steps * 1e3:.3f} ms per step incl. the epoch's host preparation ({steps} steps)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "movielens"
    if what == "movielens":
        movielens(*(int(x) for x in sys.argv[2:4]))
    else:
        synthetic(int(sys.argv[2]) if len(sys.argv) > 2 else 32,
                  tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (16, 64, 256, 1024, 4096))
