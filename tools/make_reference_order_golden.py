#!/usr/bin/env python
"""Engine-side dumps of the five MovieLens protocol cases in REFERENCE ORDER -> tests/golden/reference_order_*.npz.

What a maintainer with cargo compares the crate against (integration/rust_check/src/main.rs writes the crate's side,
tools/compare_with_crate.py reports the first divergence).  One file per case of the reference's own tests
(/root/reference/src/models/lstm.rs:450-520, ewma.rs:463-507) plus `reference_order_streams.npz` (generator streams and
the split).  Per case:

  shuffled_order        (first item, length) of the first 1 000 subsequences after `parameters.rng().shuffle` (sequence_model.rs:84)
  worker_seeds          the 16 bytes of `XorShiftRng::from_seed(parameters.rng().gen())` per worker (:97)
  first_epoch_order     per worker: (first item, length) of the first 1 000 subsequences after the first `thread_rng.shuffle` (:109)
  first_epoch_raw_draws per worker: the next 1 000 `negative_item_range.sample(thread_rng)` (:58-65, :137)
  negatives, tries      worker 0: the negatives the engine's mode chose for the first 1 000 loss terms and the draws each took
  fit_loss_lagged       what the crate's `fit` returns (:157 before :160, :173-177); fit_loss_true: the mean loss beside it
  test_ranks, test_mrr  evaluation.rs:12-48 on the protocol's test split

Produced with the CPU oracle's reference-order mode — which tests/test_parity_gpu.py compares with the ENGINE's bit for bit on
the same five cases — and cross-checked here against a replay of sequence_model.rs:76-98, :109, :137 written in Python
over sbr_rs_amd.rng (the same replay main.rs performs over `rand` 0.5): the replay and the oracle's actual run must agree on
the visiting order and on every negative, or this script fails.  tests/test_reference_order_golden.py: the oracle (CPU) and
the engine (GPU) reproduce the committed files.

    python tools/make_reference_order_golden.py            # rewrites tests/golden/reference_order_*.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import GOLDEN, LOSS_HINGE, LOSS_WARP, hparams, movielens_protocol  # noqa: E402
from sbr_rs_amd._abi import Debug, ModelKind  # noqa: E402
from sbr_rs_amd.rng import XorShiftRng  # noqa: E402

KEEP = 1000
T, DIM, EPOCHS = 128, 32, 10
CASES = [("lstm hinge 1 thread", ModelKind.LSTM_NORMAL, LOSS_HINGE, 1), ("lstm hinge 2 threads", ModelKind.LSTM_NORMAL, LOSS_HINGE, 2),
         ("lstm warp", ModelKind.LSTM_NORMAL, LOSS_WARP, 1), ("ewma hinge", ModelKind.EWMA, LOSS_HINGE, 1), ("ewma warp", ModelKind.EWMA, LOSS_WARP, 1)]


def case_file(name: str) -> str:
    return os.path.join(GOLDEN, "reference_order_" + name.replace(" ", "_") + ".npz")


def subsequences_of(train, max_len=T):
    """sequence_model.rs:76-83: chunks of every user (first chunk short, data.rs:406-431) with more than two items, in user order;
    (first item id, length, offset into item_ids)."""
    out = []
    ptr, items = train.user_pointers, train.item_ids
    for u in range(len(ptr) - 1):
        a, b = int(ptr[u]), int(ptr[u + 1])
        idx = a
        while idx < b:
            mod = (b - idx) % max_len
            cs = max_len if mod == 0 else mod
            if cs > 2:
                out.append((int(items[idx]), cs, idx))
            idx += cs
    return out


def replay(train, model_rng: XorShiftRng, threads: int, num_items: int):
    """The training driver's index work outside the model (what main.rs replays over rand 0.5)."""
    subs = subsequences_of(train)
    perm = model_rng.permutation(len(subs))          # parameters.rng().shuffle(&mut subsequences)
    subs = [subs[i] for i in perm]
    part = len(subs) // threads                        # :91; the zip at :94-98 keeps `threads` chunks
    seeds, orders, draws, rngs = [], [], [], []
    for q in range(threads):
        seed = model_rng.gen_seed()                    # XorShiftRng::from_seed(parameters.rng().gen())
        r = XorShiftRng.from_seed(seed)
        chunk = subs[q * part:(q + 1) * part]
        p2 = r.permutation(len(chunk))                 # first epoch: thread_rng.shuffle(partition)
        chunk = [chunk[i] for i in p2]
        seeds.append(np.frombuffer(seed, dtype=np.uint8).copy())
        orders.append(chunk)
        rr = r.clone()
        draws.append(np.array([rr.uniform(0, num_items) for _ in range(KEEP)], dtype=np.uint32))
        rngs.append(r)
    return subs, seeds, orders, draws


def order_array(subs):
    return np.array([(s[0], s[1]) for s in subs[:KEEP]], dtype=np.uint32).reshape(-1, 2)


def first_steps(plan, world, nmb):
    """(first item, length) per visited subsequence and worker 0's negatives / tries, from the model's own first epoch."""
    seen = [[] for _ in range(world)]
    negs, tries = [], []
    for mb in range(nmb):
        if all(len(s) >= KEEP for s in seen) and sum(map(len, negs)) >= KEEP:
            break
        plan.step(mb)
        for q in range(world):
            rows = plan.minibatch_rows(mb, q) if world > 1 else plan.minibatch_rows(mb)
            fetch = (lambda w: plan.debug_fetch(w, rows, q)) if world > 1 else (lambda w: plan.debug_fetch(w, rows))
            seen[q].append((int(fetch(Debug.IN_IDX)[0]), rows + 1))
            if q == 0:
                negs.append(fetch(Debug.NEGATIVES).copy())
                tries.append(fetch(Debug.TRIES).copy())
    return seen, np.concatenate(negs)[:KEEP].astype(np.uint32), np.concatenate(tries)[:KEEP].astype(np.uint32)


def build_case(name, kind, loss, threads, make_model):
    """make_model(hp) -> a model with the oracle's / engine's Python surface, already in reference order."""
    data, train, test, rng = movielens_protocol()
    hp = hparams(data.num_items(), T, DIM, int(kind), loss, epochs=EPOCHS, B=1, seed=rng.state_seed(), ndev=threads)
    m = make_model(hp)
    subs, seeds, orders, draws = replay(train, XorShiftRng.from_seed(m.get_rng()), threads, data.num_items())
    plan = m.fit_begin(train.user_pointers, train.item_ids)
    seen, negs, tries = first_steps(plan, threads, plan.epoch_prepare())
    plan.close()
    # the replay IS what the model does: visiting order of every worker, and worker 0's negatives out of its raw draws
    for q in range(threads):
        want = [(s[0], s[1]) for s in orders[q][:len(seen[q])]]
        assert seen[q][:KEEP] == want[:KEEP], f"{name}: worker {q} visits its partition in another order than the replay"
    pos = np.cumsum(tries) - 1
    ok = pos < KEEP
    assert np.array_equal(negs[ok], draws[0][pos[ok]]), f"{name}: the chosen negatives are not the replayed draws"
    m2 = make_model(hp)  # the whole protocol run on a fresh model
    loss_true = m2.fit(train.user_pointers, train.item_ids)
    mrr, ranks = m2.mrr_score(test.user_pointers, test.item_ids)
    return dict(shuffled_order=order_array(subs), worker_seeds=np.stack(seeds), first_epoch_order=np.stack([order_array(o) for o in orders]),
                first_epoch_raw_draws=np.stack(draws), negatives=negs, tries=tries, fit_loss_lagged=np.float32(m2.last_fit_lagged_loss()),
                fit_loss_true=np.float32(loss_true), test_ranks=np.asarray(ranks, dtype=np.uint32), test_mrr=np.float32(mrr),
                num_subsequences=np.int64(len(subs)))


def streams():
    """The generator streams main.rs dumps (`Streams`), from the Python statement of rand 0.5 and the oracle's normal sampler."""
    import ctypes as C

    from oracle.oracle import lib

    seed = bytes([42] * 16)
    a = XorShiftRng.from_seed(seed)
    out = {"next_u32": np.array([a.next_u32() for _ in range(8)], dtype=np.uint32)}
    b = XorShiftRng.from_seed(seed)
    out["uniform_u64"] = np.array([b.uniform(0, (1 << 64) - 1) for _ in range(2)], dtype=np.uint64)
    c = XorShiftRng.from_seed(seed)
    out["uniform_usize_1683"] = np.array([c.uniform(0, 1683) for _ in range(16)], dtype=np.uint64)
    out["shuffle_10"] = XorShiftRng.from_seed(seed).permutation(10).astype(np.uint64)
    normal = np.zeros(8, dtype=np.float64)  # orc_rand_stream what = 2: Normal(mean a, std b) as f64 — see tests/test_rand05.py
    s = np.frombuffer(seed, dtype=np.uint8).copy()
    lib().orc_rand_stream(s.ctypes.data_as(C.c_void_p), 2, 0, 0, normal.ctypes.data_as(C.c_void_p), 8)
    out["normal_bits"] = normal.view(np.uint64)
    out["gen_seed16"] = np.frombuffer(XorShiftRng.from_seed(seed).gen_seed(), dtype=np.uint8).copy()
    data, train, test, _rng = movielens_protocol()

    def fnv(items):
        h = 0xCBF29CE484222325
        for x in items:
            for byte in int(x).to_bytes(8, "little"):
                h = ((h ^ byte) * 0x100000001B3) & ((1 << 64) - 1)
        return h

    for nm, m in (("train", train), ("test", test)):
        lens = np.diff(m.user_pointers.astype(np.int64))
        out[f"{nm}_users_with_data"] = np.int64((lens > 0).sum())
        out[f"{nm}_interactions"] = np.int64(len(m.item_ids))
        out[f"{nm}_items_fnv"] = np.uint64(fnv(m.item_ids))
    return out


def main():
    from oracle.oracle import OracleModel

    def make(hp):
        m = OracleModel(hp)
        m.set_reference_order(True)
        return m

    np.savez(os.path.join(GOLDEN, "reference_order_streams.npz"), **streams())
    for name, kind, loss, threads in CASES:
        d = build_case(name, kind, loss, threads, make)
        np.savez(case_file(name), **d)
        print(f"{name}: {int(d['num_subsequences'])} subsequences, fit (crate's figure) {float(d['fit_loss_lagged']):.6f}, test MRR {float(d['test_mrr']):.4f}")


if __name__ == "__main__":
    main()
