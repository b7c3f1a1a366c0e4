"""Oracle parity of ONE optimiser step at sizes the oracle cannot run whole (the bench's own regime), by sampling.

The side under test ("full": the HIP engine on the GPU; in the CPU self-test a second oracle) runs the first optimiser step of
a fit on the whole minibatch.  The oracle then
  1. runs forward + negative sampling + loss + BPTT for a SAMPLE of the minibatch's sequences only — with the epoch key and
     the position counters those sequences have in the full minibatch (oracle.step_local_sample; a sequence's rows depend on
     the parameters and on its own items only) — and every packed row of the sample is compared bit for bit with the same
     row of the full step: indices, hidden states, negatives, trip counts, coefficients, losses, dX and (LSTM) dZ;
  2. recomputes, for a sample of item-table rows, the row's optimiser step from EVERY entry of the full minibatch that touches
     it, in the contract's order (oracle.row_step: (packed row, kind) order, 256-entry chunks, Adagrad with L2), from the full
     step's own H / dX / coef — and compares E, E_acc, b, b_acc of those rows after the step bit for bit;
  3. (LSTM) recomputes sampled elements of the dense gradient as the contract's chunked fma chain over ALL packed rows
     (oracle.dense_chain) from the full step's X / H / dZ columns, compares them with the step's dense-gradient block, and
     compares W, W_acc, bW, bW_acc after the step with the oracle's dense Adagrad applied to that block.
What is not re-derived by the oracle: H / dX / dZ of the unsampled sequences (inputs of 2 and 3 are the full step's own
arrays, verified on the sample), and EWMA's d-element alpha gradient.
≙ one iteration of /root/reference/src/models/sequence_model.rs:111-169 for a minibatch."""
from __future__ import annotations

import numpy as np

from sbr_rs_amd._abi import Debug, Param


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.flatnonzero((_bits(a) if a.dtype == np.float32 else a).ravel() != (_bits(b) if b.dtype == np.float32 else b).ravel())
    assert bad.size == 0, f"{what}: {bad.size}/{a.size} differ; first at {bad[0]}: full={a.ravel()[bad[0]]!r} oracle={b.ravel()[bad[0]]!r}"


def count_subsequences(ptr, T):
    """Chunks of at most T items with more than two items (data.rs:406-431 + sequence_model.rs:81)."""
    lens = np.diff(np.asarray(ptr, dtype=np.int64))
    rem = lens % T
    first = np.where(rem == 0, np.minimum(lens, T), rem)  # the short chunk comes first
    return int(np.sum(first > 2) + np.sum((lens - first) // T))


def pick_sequences(nb, nsel, rs):
    """Ascending packed-order indices across the first, middle and last tiles of the minibatch (tile = 16 / 32 / 64 sequences;
    beyond tile 1 024 of 32 when the minibatch is that large: the unfolded tile order of the sequence-resident kernels)."""
    fixed = [0, 1, 15, 16, 31, 32, 63, 64, nb // 2, nb // 2 + 1, nb - 65, nb - 33, nb - 17, nb - 2, nb - 1, 32 * 1024 + 5, 32 * 1100 + 31]
    fixed = [b for b in fixed if 0 <= b < nb]
    extra = rs.choice(nb, size=max(0, min(nb, nsel) - len(set(fixed))), replace=False) if nb > len(set(fixed)) else []
    return np.unique(np.concatenate([np.array(fixed, dtype=np.int64), np.asarray(extra, dtype=np.int64)])).astype(np.uint32)


def check_first_step(full, o, po, *, lstm: bool, nb: int, nsel=64, nrows=256, ndense=48, seed=0):
    """`full`: adapter of the side that runs the whole step — .rows (packed rows of minibatch 0), .step_local(), .fetch(which),
    .apply(), .model (get_param / get_param_rows).  `o`, `po`: an oracle model with the SAME hyper-parameters (and so the same
    initial parameters) and its plan, epoch prepared, not stepped."""
    rs = np.random.RandomState(seed)
    g = full.model
    d = g.storage_dim
    I = int(g.hp.num_items)
    ng = {0: 4, 1: 3, 2: 0}[int(g.hp.model)]
    full.step_local()
    R = full.rows
    # ---- 1. sampled sequences ---------------------------------------------------------------------------------------
    sel = pick_sequences(nb, nsel, rs)
    idx, off = po.step_local_sample(0, sel)
    n = idx.size
    assert n > 0 and idx.max() < R
    whichs = [Debug.IN_IDX, Debug.OUT_IDX, Debug.HIDDEN, Debug.NEGATIVES, Debug.TRIES, Debug.COEF, Debug.LOSS, Debug.DINPUT] + ([Debug.DZ] if lstm else [])
    arrays = {}
    for w in whichs:
        arrays[w] = full.fetch(w)
        _same(arrays[w][idx], po.debug_fetch(w, n), f"sampled sequences: {w.name}")
    in_idx, out_idx, neg, coef, H, dX = (arrays[w] for w in (Debug.IN_IDX, Debug.OUT_IDX, Debug.NEGATIVES, Debug.COEF, Debug.HIDDEN, Debug.DINPUT))
    assert off[-1] <= R and int(off[1]) == nb
    # ---- 3a. sampled elements of the dense gradient (before the step is applied: it reads the initial table) -----------
    dense = full.fetch(Debug.DENSE_GRAD) if lstm else None
    if lstm:
        assert d == g.dim, "the dense-gradient sample is written for the kernels' own widths"
        nz = ng * d
        dZ = arrays[Debug.DZ]
        E0 = g.get_param(Param.ITEM_EMBEDDING).reshape(I, d)  # the initial table (LSTM configurations: 1e6 items = 0.5 GB)
        offi = off.astype(np.int64)
        rr = np.arange(R, dtype=np.int64)
        row_t = np.searchsorted(offi, rr, side="right") - 1
        prev = np.where(row_t > 0, offi[np.maximum(row_t - 1, 0)] + (rr - offi[row_t]), -1)
        from oracle.oracle import dense_chain

        third = ndense // 3
        ks = np.concatenate([rs.randint(0, d, third), d + rs.randint(0, d, third), np.full(ndense - 2 * third, 2 * d)])
        js = rs.randint(0, nz, ks.size)
        for k, j in zip(ks, js):
            dzc = np.ascontiguousarray(dZ[:, j])
            if k < d:
                a = E0[in_idx, k]                                                       # x_t = E[in_t]
            elif k < 2 * d:
                a = np.where(prev >= 0, H[np.maximum(prev, 0), k - d], np.float32(0.0))  # h_{t-1}, 0 at t = 0
            else:
                a = None                                                                # bias row: plain add chain
            want = dense_chain(None if a is None else a.astype(np.float32), dzc)
            _same(np.array([dense[k * nz + j]], dtype=np.float32), np.array([want], dtype=np.float32), f"dense gradient element ({k}, {j})")
        del E0
    # ---- 2. sampled item rows: the whole optimiser step of the row ------------------------------------------------------
    all_rows = np.stack([in_idx, out_idx, neg], axis=1).ravel()  # entry e = 3 r + kind
    order = np.argsort(all_rows, kind="stable")                  # (row, packed row, kind) order
    sorted_rows = all_rows[order]
    pick = rs.choice(R, size=min(R, nrows), replace=False)
    items = np.unique(np.concatenate([in_idx[pick[: nrows // 3]], out_idx[pick[nrows // 3: 2 * nrows // 3]], neg[pick[2 * nrows // 3:]]])).astype(np.uint32)
    before = {p: g.get_param_rows(p, items) for p in (Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.ITEM_BIAS_ACC)}
    for p in before:  # the full side's initial rows are the oracle's
        _same(before[p], o.get_param_rows(p, items), f"initial {p.name} rows")
    full.apply()
    after = {p: g.get_param_rows(p, items) for p in before}
    lo = np.searchsorted(sorted_rows, items, side="left")
    hi = np.searchsorted(sorted_rows, items, side="right")
    max_entries = 0
    for i, item in enumerate(items):
        e = order[lo[i]:hi[i]]
        assert e.size > 0
        max_entries = max(max_entries, e.size)
        r, kind = e // 3, e % 3
        vecs = np.where((kind == 0)[:, None], dX[r], H[r]).astype(np.float32)
        scale = np.where(kind == 0, np.float32(1.0), np.where(kind == 1, -coef[r], coef[r])).astype(np.float32)
        pad = lambda v: np.concatenate([v, np.zeros(d - v.size, dtype=np.float32)]) if v.size < d else v
        w, acc, b, bacc = o.row_step(vecs, scale, kind != 0, pad(before[Param.ITEM_EMBEDDING][i]), pad(before[Param.ITEM_EMBEDDING_ACC][i]),
                                     before[Param.ITEM_BIAS][i], before[Param.ITEM_BIAS_ACC][i])
        dl = g.dim
        _same(after[Param.ITEM_EMBEDDING][i], w[:dl], f"item {item}: embedding row after the step ({e.size} entries)")
        _same(after[Param.ITEM_EMBEDDING_ACC][i], acc[:dl], f"item {item}: accumulator row after the step")
        _same(np.array([after[Param.ITEM_BIAS][i]]), np.array([b]), f"item {item}: bias after the step")
        _same(np.array([after[Param.ITEM_BIAS_ACC][i]]), np.array([bacc]), f"item {item}: bias accumulator after the step")
    # ---- 3b. dense parameters after the step ------------------------------------------------------------------------------
    if lstm and d == g.dim:
        o.apply_dense(dense)
        for p in (Param.LSTM_W, Param.LSTM_W_ACC, Param.LSTM_B, Param.LSTM_B_ACC):
            _same(g.get_param(p), o.get_param(p), f"{p.name} after the step")
    return {"rows": int(R), "sampled_sequences": int(sel.size), "sampled_rows": int(n), "sampled_items": int(items.size), "max_entries_per_item": int(max_entries)}


def nearest_touched(sorted_touched, boundaries):
    """For every row index B in `boundaries`: the nearest touched row below B and the nearest at or above it (a touched row
    on either side of every ownership change of a partitioned table)."""
    out = []
    for B in boundaries:
        i = int(np.searchsorted(sorted_touched, B, side="left"))
        if i > 0:
            out.append(int(sorted_touched[i - 1]))
        if i < sorted_touched.size:
            out.append(int(sorted_touched[i]))
    return np.unique(np.array(out, dtype=np.int64))


def check_first_step_multi(full, o, po, *, lstm: bool, nb: int, nsel=24, nrows=160, seed=0, boundaries=()):
    """The multi-device step (user-sharded data parallelism) at a size the oracle cannot run whole.
    `full`: adapter of the side that runs `world` devices' first step — .world, .rows(q), .step_local_all(), .fetch(q, which),
    .apply_all(), .model(q).  `o`, `po`: an oracle model with num_devices = world and its plan, epoch prepared, not stepped.
      1. per device: a sample of its sequences against the oracle's restatement of that device's half-step (its partition, its
         negative-draw key);
      2. sampled item rows: every device's own entries reduced in the contract's order (oracle.row_reduce), the devices' sums added
         in device order (the first toucher initialises), ONE optimiser update (oracle.row_apply) — against the row on the first
         and on the last replica after the step;
      3. (LSTM) the devices' dense-gradient blocks added in device order, the oracle's dense update of that sum — against the dense
         parameters of the first and the last replica.
    `boundaries` (a partitioned item table, BASELINE configs[4]): row indices at which the owner of a row changes — logical slice
    starts and the physical page runs' first rows; the nearest touched row on each side of every one of them, the last touched rows
    of the table (the uneven last slice) and a sample of rows that several devices touch join the sampled rows."""
    from oracle.oracle import row_reduce

    rs = np.random.RandomState(seed)
    world = full.world
    g0 = full.model(0)
    d = g0.storage_dim
    assert d == g0.dim
    full.step_local_all()
    per = []
    for q in range(world):
        R = full.rows(q)
        sel = pick_sequences(nb, nsel, rs)
        idx, _off = po.step_local_sample(0, sel, device=q)
        n = idx.size
        assert n > 0 and idx.max() < R
        arr = {}
        for w in [Debug.IN_IDX, Debug.OUT_IDX, Debug.HIDDEN, Debug.NEGATIVES, Debug.TRIES, Debug.COEF, Debug.LOSS, Debug.DINPUT]:
            arr[w] = full.fetch(q, w)
            _same(arr[w][idx], po.debug_fetch(w, n, device=q), f"device {q}: sampled sequences: {w.name}")
        all_rows = np.stack([arr[Debug.IN_IDX], arr[Debug.OUT_IDX], arr[Debug.NEGATIVES]], axis=1).ravel()
        order = np.argsort(all_rows, kind="stable")
        per.append({"H": arr[Debug.HIDDEN], "dX": arr[Debug.DINPUT], "coef": arr[Debug.COEF], "all": all_rows, "order": order,
                    "sorted": all_rows[order], "dense": full.fetch(q, Debug.DENSE_GRAD) if lstm else None})
    # sampled items: touched by some device (a third each from inputs, targets, negatives of random devices)
    picks = []
    for k in range(3):
        q = int(rs.randint(world))
        rr = rs.choice(per[q]["all"].size // 3, size=nrows // 3, replace=False)
        picks.append(per[q]["all"][3 * rr + k])
    near = np.zeros(0, dtype=np.int64)
    if len(boundaries):  # a partitioned table: touched rows on both sides of every ownership change, and the last rows of the table
        touched = np.unique(np.concatenate([P["sorted"] for P in per]))
        near = nearest_touched(touched, list(boundaries))
        picks.append(near.astype(picks[0].dtype))
        picks.append(touched[-2:].astype(picks[0].dtype))
        several = None  # rows that more than one device touches: the device-ordered merge at the owner
        for q in range(1, world):
            both = np.intersect1d(per[0]["sorted"], per[q]["sorted"])
            several = both if several is None else np.union1d(several, both)
        if several is not None and several.size:
            picks.append(several[rs.choice(several.size, size=min(32, several.size), replace=False)].astype(picks[0].dtype))
    items = np.unique(np.concatenate(picks)).astype(np.uint32)
    params = (Param.ITEM_EMBEDDING, Param.ITEM_EMBEDDING_ACC, Param.ITEM_BIAS, Param.ITEM_BIAS_ACC)
    before = {p: g0.get_param_rows(p, items) for p in params}
    for p in params:
        _same(before[p], o.get_param_rows(p, items), f"initial {p.name} rows")
    full.apply_all()
    touched_by = np.zeros(items.size, dtype=np.int64)
    for i, item in enumerate(items):
        g = gb = None
        has_b = False
        for q in range(world):
            P = per[q]
            lo, hi = np.searchsorted(P["sorted"], item, side="left"), np.searchsorted(P["sorted"], item, side="right")
            if hi == lo:
                continue
            touched_by[i] += 1
            e = P["order"][lo:hi]
            r, kind = e // 3, e % 3
            vecs = np.where((kind == 0)[:, None], P["dX"][r], P["H"][r]).astype(np.float32)
            scale = np.where(kind == 0, np.float32(1.0), np.where(kind == 1, -P["coef"][r], P["coef"][r])).astype(np.float32)
            gq, gbq, hbq = row_reduce(vecs, scale, kind != 0)
            g = gq if g is None else (g + gq).astype(np.float32)   # device order; the first toucher initialises
            if hbq:
                gb = gbq if not has_b else np.float32(gb + gbq)
                has_b = True
        assert g is not None
        w, acc, b, bacc = o.row_apply(g, gb if has_b else np.float32(0), has_b, before[Param.ITEM_EMBEDDING][i], before[Param.ITEM_EMBEDDING_ACC][i],
                                      before[Param.ITEM_BIAS][i], before[Param.ITEM_BIAS_ACC][i])
        for q in (0, world - 1):
            m = full.model(q)
            _same(m.get_param_rows(Param.ITEM_EMBEDDING, [item])[0], w, f"item {item} on replica {q}: embedding row after the step")
            _same(m.get_param_rows(Param.ITEM_EMBEDDING_ACC, [item])[0], acc, f"item {item} on replica {q}: accumulator row")
            _same(m.get_param_rows(Param.ITEM_BIAS, [item]), np.array([b]), f"item {item} on replica {q}: bias")
            _same(m.get_param_rows(Param.ITEM_BIAS_ACC, [item]), np.array([bacc]), f"item {item} on replica {q}: bias accumulator")
    if lstm:
        dense = per[0]["dense"].copy()
        for q in range(1, world):
            dense = (dense + per[q]["dense"]).astype(np.float32)
        o.apply_dense(dense)
        for q in (0, world - 1):
            for p in (Param.LSTM_W, Param.LSTM_W_ACC, Param.LSTM_B, Param.LSTM_B_ACC):
                _same(full.model(q).get_param(p), o.get_param(p), f"{p.name} on replica {q} after the step")
    return {"world": world, "sampled_items": int(items.size), "items_touched_by_several_devices": int(np.sum(touched_by > 1)),
            "rows_per_device": [int(full.rows(q)) for q in range(world)], "boundary_items": [int(x) for x in near]}
