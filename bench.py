#!/usr/bin/env python
"""bench.py — train interactions/sec of the sequence-recommender hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU.  Under a launcher — WORLD_SIZE set, e.g. the driver's `python -m torch.distributed.run` —
    this process IS a rank; without one it starts the N ranks itself through the same launcher on 127.0.0.1.)

A *step* is one optimiser step of the hot path: one minibatch of `--batch-sequences` synthetic
subsequences per GPU through forward (embedding gather + LSTM), WARP negative sampling + loss,
BPTT and the Adagrad dense + sparse update (fit_sequence_model's inner loop,
/root/reference/src/models/sequence_model.rs:111-169).  An *interaction* is one (input, target)
pair = one loss term (`examples += n-1`, sequence_model.rs:158).

Workload (default, N = 1): BASELINE.json configs[2] — synthetic 100K users x 1M items, seq_len <= 64,
embedding_dim 128, LSTM(Normal) + WARP + Adagrad, the largest single-GPU configuration and the one
the north-star HBM-roofline target is quoted on.  (configs[1], MovieLens-100K, is 1.3K
subsequences — a parity/MRR case, run here untimed for the `test_mrr` field and in
tests/test_parity_gpu.py.)  With N > 1 GPUs the default is BASELINE.json configs[3]'s per-GPU shape —
125 000 users per GPU (1M users at N = 8), seq_len <= 128, the same model — users sharded, every GPU
on its own partition, and per step one all-to-all + one all-gather (RCCL) of dense per-owner gradient
chunks realise the synchronised optimiser step; `--partition-table --model ewma --loss
hinge --dim 256 --items 10000000` is configs[4].  `--users` / `--max-len` override either default.
`--simulate-world N` (one GPU, no process group) runs rank 0's share of an N-GPU step with the real
scatter / owner-reduce / table-update kernels and device copies in place of the collectives, and
prints the kernel-side cost of the exchange and the bytes every xGMI link would carry.

Prints ONE JSON line (rank 0, the last line of stdout).  `roofline` describes the gather +
WARP-score kernel against the HBM roofline (`traffic` = PMC bytes measured at this operating point,
profiles/score_kernel_traffic.json + profiles/r03_counter_calibration.md, else null);
`kernels` lists every kernel family's time inside the timed region (HIP events on the engine's
streams; concurrent families include each other's contention); `kernels_standalone` comes from a
second, UNTIMED pass with stream overlap off (N = 1): every family alone, with the sparse update's
HBM figure and the GEMM-shaped kernels' MFMA figures; `roofline_mfma` are the in-region MFMA
figures; `cpu_baseline` is the CPU oracle (a scalar C port of the same algorithm; the Rust
reference cannot be built here) timed on the host in the reference's parallel shape — every core a
worker on ONE shared parameter set, Hogwild and synchronised — and single-thread, each leg bounded in
wall time, rank 0, N = 1 only; `test_mrr` is configs[1]
(MovieLens-100K), untimed.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290
FP32_MFMA_PEAK_TF = 157.3  # dense f32-input MFMA peak


def synthetic_csr(num_users: int, num_items: int, max_len: int, seed: int = 42, zipf: bool = False):
    """BASELINE.md §3 generator: len_u ~ U{3..max_len}, timestamps = position; items ~ U[0, num_items)
    (the pure-roofline variant: no cache reuse) or Zipf(1.0) over a random permutation of the
    catalogue (the realistic variant)."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(3, max_len + 1, size=num_users).astype(np.uint64)
    ptr = np.zeros(num_users + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum(lens)
    nnz = int(ptr[-1])
    if zipf:
        cdf = np.cumsum(1.0 / np.arange(1, num_items + 1, dtype=np.float64))
        ranks = np.searchsorted(cdf, rs.random_sample(nnz) * cdf[-1])
        items = rs.permutation(num_items)[np.minimum(ranks, num_items - 1)].astype(np.uint32)
    else:
        items = rs.randint(0, num_items, size=nnz).astype(np.uint32)
    return ptr, items


def make_hp(args, world, rank, model_kind, loss, num_items, epochs=1, batch=None, dim=None, max_len=None):
    from sbr_rs_amd._abi import make_hparams

    par = 0 if getattr(args, "parallelism", "sync") == "async" else 1
    return make_hparams(num_items, max_len or args.max_len, dim or args.dim, 0.16, 0.0004, model_kind, loss, 0, par,
                        bytes([42] * 16), epochs, world, rank, batch or args.batch_sequences)


def measured_ceiling(row_bytes: int, cached_table: bool):
    """Random-row gather rate this device sustains (tools/hbm_ceiling.hip, profiles/r02_hbm_ceiling.jsonl): the
    figure the score kernel's REAL traffic is to be held against (the roofline fractions use the 8 TB/s spec)."""
    path = os.path.join(ROOT, "profiles", "r02_hbm_ceiling.jsonl")
    best = None
    try:
        for line in open(path):
            r = json.loads(line)
            if r.get("access") != "random row gather" or r.get("row_bytes") != row_bytes:
                continue
            if (r.get("table_GiB", 0) < 1.0) != cached_table:
                continue
            best = max(best or 0.0, float(r["GBps"]))
    except Exception:
        return None
    return best


def measured_traffic(which, rows_per_launch, k_mean, d, items=None):
    """(upper, lower) HBM bytes per launch of the score kernel from profiles/score_kernel_traffic.json if a profile was taken
    at this run's operating point (rows per launch and negatives per row within 5 %, same table size), else (None, None).  The
    file holds one entry per profiled operating point under "warm" / "cold" (a list, or round 3's single entry)."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "score_kernel_traffic.json")))[which]
    except Exception:
        return None, None
    for e in (prof if isinstance(prof, list) else [prof]):
        if d != e.get("dim", 128) or (items is not None and e.get("items") not in (None, items)):
            continue
        if abs(e["rows_per_launch"] / max(rows_per_launch, 1) - 1) > 0.05 or abs(e["mean_negatives_scored"] / max(k_mean, 1e-9) - 1) > 0.05:
            continue
        return e["hbm_bytes_per_launch"], e["hbm_bytes_per_launch_lower"]
    return None, None


def live_traffic(kernel, warmup_dispatches, timed_dispatches, timeout_s=150):
    """HBM bytes per launch of `kernel` at THIS run's operating point, measured now: two rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, only --kernel-trace beside them — MI355X_MICROARCH.md "HBM") around a child run of this
    command line (same workload, batch, steps; extras off), the timed dispatches only.  Corrections as calibrated in
    profiles/r03_counter_calibration.md: both counters in KiB; FETCH_SIZE counts half of every coalesced read and of 512-byte
    row gathers; an isolated 4-byte bias read may be counted as a whole 64-byte sector (not halved): upper = (2 F + W) KiB,
    lower = upper - 64 B x (1 + k) x rows.  Returns a dict, or None if rocprofv3 is missing / a pass fails."""
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_dispatches import per_dispatch, step_window

    tmp = tempfile.mkdtemp(prefix="sbr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    drop = {"--traffic"}
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a in drop:
            skip = True
            continue
        if a.startswith("--traffic="):
            continue
        argv.append(a)
    child = [sys.executable, os.path.abspath(__file__)] + argv + ["--traffic", "off", "--no-cpu-baseline", "--no-mrr", "--batch-sweep=",
                                                                  "--standalone-steps", "0", "--cold-items", "0"]
    vals, line, step = {}, None, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            res = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "run", "--"] + child, cwd="/tmp", env=env,
                                 capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if res.returncode != 0 or not dbs:
                return None
            # the timed region's dispatches only (the child goes on with the second, all-timers pass and the model keeps learning:
            # later launches try more negatives per row)
            vals[counter] = per_dispatch(dbs[0], counter, kernel)[warmup_dispatches:warmup_dispatches + timed_dispatches]
            step[counter] = step_window(dbs[0], counter, kernel, warmup_dispatches, timed_dispatches)  # every kernel of the timed steps
            line = json.loads([x for x in res.stdout.splitlines() if x.startswith('{"metric"')][-1])
        n = min(len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"]))
        if n == 0 or not line or not line.get("roofline"):
            return None
        f, w = sum(vals["FETCH_SIZE"][:n]) / n, sum(vals["WRITE_SIZE"][:n]) / n
        roof = line["roofline"]
        up = (2.0 * f + w) * 1024.0
        whole = None
        if step.get("FETCH_SIZE") and step.get("WRITE_SIZE"):
            names = sorted(set(step["FETCH_SIZE"]) | set(step["WRITE_SIZE"]))
            by = {k: (2.0 * step["FETCH_SIZE"].get(k, 0.0) + step["WRITE_SIZE"].get(k, 0.0)) * 1024.0 / timed_dispatches for k in names}
            whole = {"hbm_bytes_per_step": sum(by.values()), "by_kernel": {k: v for k, v in sorted(by.items(), key=lambda kv: -kv[1]) if v >= 1e6},
                     "read_bytes_per_step": sum(step["FETCH_SIZE"].values()) * 2048.0 / timed_dispatches,
                     "written_bytes_per_step": sum(step["WRITE_SIZE"].values()) * 1024.0 / timed_dispatches,
                     "what": "every dispatch between the first and the last timed dispatch of the roofline kernel, (2 F + W) x 1024 per step"}
        return {"step": whole, "hbm_bytes_per_launch": up, "hbm_bytes_per_launch_lower": up - 64.0 * (1 + roof["mean_negatives_scored"]) * roof["rows_per_launch"],
                "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "dispatches_averaged": n, "rows_per_launch": roof["rows_per_launch"],
                "mean_negatives_scored": roof["mean_negatives_scored"], "kernel": kernel}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(args, model_kind=0, loss_kind=2):
    """The oracle (kind "port") in the REFERENCE'S PARALLEL SHAPE on the host (sequence_model.rs:90-102): the subsequences of a
    bounded sample of the same generator are cut into one partition per worker thread, every worker runs on ONE shared
    parameter set (Arc<HogwildParameter>, lstm.rs:175-181) and takes one optimiser step per subsequence (the reference's
    schedule, :111-169).  Workers = every host core (the reference's default, lstm.rs:68); both of its parallelism modes are
    timed — Asynchronous = Hogwild, no locks (mod.rs:36-38), and Synchronous = rendezvous + one update at a time
    (mod.rs:39-40, sequence_model.rs:163-166) — beside the single-thread figure (num_threads(1), lstm.rs:462) and two
    intermediate Hogwild worker counts (the scaling curve: every worker's dense Adagrad step rewrites the same 1 MB of LSTM weights,
    so the shared-parameter shape stops scaling long before the core count).  Each leg is bounded by --cpu-seconds of wall time;
    `value` is the FASTEST leg of all (`mode` names it, `cores` is its worker count) — more threads being slower is a property of
    this port, not of the reference.  Thread timing orders the updates: a throughput baseline, not a parity run."""
    from oracle.oracle import OracleModel

    users = min(args.cpu_users, args.users)
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, args.cpu_threads if args.cpu_threads > 0 else cores))
    ptr, items = synthetic_csr(users, args.items, args.max_len, seed=43)
    hp = make_hp(args, 1, 0, model_kind, loss_kind, args.items, epochs=1_000_000, batch=1)  # bounded by time, not by epochs
    m = OracleModel(hp)  # ONE model: the three legs continue training the same shared parameters
    legs = {}
    plan = [("single_thread", 1, True)] + [(f"hogwild_{w}_threads", w, False) for w in (16, 64) if w < workers] + \
           [("all_cores_hogwild", workers, False), ("all_cores_synchronous", workers, True)]
    for name, w, sync in plan:
        rows, secs, _loss = m.fit_threads(ptr, items, w, sync, args.cpu_seconds)
        legs[name] = {"interactions_per_s": rows / secs, "interactions": rows, "seconds": secs, "workers": w}
    best = max(legs, key=lambda k: legs[k]["interactions_per_s"])  # the strongest leg of all, whatever its worker count
    out = {"value": legs[best]["interactions_per_s"], "unit": "interactions/s", "cores": legs[best]["workers"], "kind": "port", "mode": best,
           "sample": f"{users} users of the same generator, {args.model}+{args.loss} dim {args.dim}, {args.items} items; {legs[best]['workers']} worker thread(s) (the fastest "
                     f"of the legs timed with 1 .. {workers} workers) on ONE shared parameter set, one partition each (sequence_model.rs:90-102), one "
                     f"optimiser step per subsequence; "
                     f"each leg bounded by {args.cpu_seconds:.0f} s of wall time; C oracle",
           "single_thread_value": legs["single_thread"]["interactions_per_s"], "host_cores_available": cores, "legs": legs}
    del m
    out["rust_toolchain"] = rust_toolchain_probe()
    if workload_label(args, 1).startswith("BASELINE.json configs[2]"):
        out["movielens_batch1"] = cpu_movielens_batch1(workers)
    return out


def rust_toolchain_probe():
    """BASELINE.md section 2.2 "probe, do not assume": is there a Rust toolchain on THIS host?  The reference's CPU path is Rust +
    rayon; where `cargo` exists and SBR_RS_CHECKOUT names a checkout of maciejkula/sbr-rs whose dependencies are already
    fetched, its own Criterion bench (benches/benchmark.rs) is run (offline, bounded) and its output kept — that would be a
    "reference" CPU figure.  Everywhere else the C port above stands in, and this says why."""
    import shutil
    import subprocess

    cargo, rustc = shutil.which("cargo"), shutil.which("rustc")
    out = {"cargo": cargo, "rustc": rustc, "status": "present" if cargo and rustc else "absent"}
    if not (cargo and rustc):
        out["consequence"] = "cpu_baseline.kind stays \"port\" (the C oracle); integration/rust_check/ is the comparison kit for a host that has cargo"
        return out
    src = os.environ.get("SBR_RS_CHECKOUT")
    if not src or not os.path.exists(os.path.join(src, "Cargo.toml")):
        out["consequence"] = "toolchain present but no crate checkout (SBR_RS_CHECKOUT unset; no network here): cargo bench not run"
        return out
    try:
        res = subprocess.run([cargo, "bench", "--offline", "--bench", "benchmark"], cwd=src, capture_output=True, text=True, timeout=900)
        out["cargo_bench_rc"] = res.returncode
        out["cargo_bench_tail"] = (res.stdout + res.stderr)[-1500:]
    except Exception as e:
        out["cargo_bench_error"] = repr(e)
    return out


def cpu_movielens_batch1(workers):
    """The CPU leg of `test_mrr.batch_sequences_1`: BASELINE.json configs[1] (MovieLens-100K, LSTM Normal, dim 32, WARP,
    Adagrad, 10 epochs) at one subsequence per optimiser step — the reference's own schedule — through the C oracle: on ONE
    host core (the sequential contract run, whose result the GPU reproduces bit for bit), and in the reference's parallel shape
    on every core (one shared parameter set, Hogwild and synchronised; fit time only)."""
    from helpers import movielens_protocol
    from oracle.oracle import OracleModel
    from sbr_rs_amd._abi import make_hparams

    data, train, _test, rng = movielens_protocol()
    mk = lambda: OracleModel(make_hparams(data.num_items(), 128, 32, 0.16, 0.0004, 0, 2, 0, 1, rng.state_seed(), 10, 1, 0, 1))
    m = mk()
    t0 = time.perf_counter()
    loss = m.fit(train.user_pointers, train.item_ids)
    dt = time.perf_counter() - t0
    out = {"fit_seconds": dt, "fit_loss": loss, "cores": 1, "kind": "port",
           "train_interactions_per_s": 10 * len(train.item_ids) / dt,
           "config": "MovieLens-100K, LSTM Normal, dim 32, WARP, Adagrad, 10 epochs, max_sequence_length 128, batch_sequences 1"}
    w = min(workers, 256)
    for name, sync in (("all_cores_hogwild", False), ("all_cores_synchronous", True)):
        try:
            rows, secs, lv = mk().fit_threads(train.user_pointers, train.item_ids, w, sync, 0.0)
            out[name] = {"fit_seconds": secs, "fit_loss": lv, "workers": w, "interactions": rows}
        except Exception as e:  # fewer subsequences than workers: the reference panics there (chunks_mut(0))
            out[name] = {"error": repr(e), "workers": w}
    return out


def reference_criterion_bench(samples: int = 5):
    """benches/benchmark.rs:16-71 through the Python host mirror (tools/criterion_bench.py): `fit` on 10 000 sampled MovieLens-100K
    interactions, max_sequence_length 128, dim 32, hinge, Adagrad, three epochs per call, re-fitting one model; the reference's
    schedule (one subsequence per optimiser step) and 16 subsequences per step; the C oracle's time for the same call on one host
    core beside it (profiles/r04_criterion_bench.md)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import criterion_bench as cb
    from oracle.oracle import OracleModel

    data = cb.sample_data(10_000)
    out = {"config": "benches/benchmark.rs: fit on 10 000 sampled MovieLens-100K interactions, max_sequence_length 128, dim 32, hinge, Adagrad, "
                     "3 epochs per call, 1 worker; ms per fit call, mean of %d calls on one model after a warm-up call" % samples}
    for kind in ("lstm", "ewma"):
        for batch in (1, 16):
            model = cb.build(kind, data.num_items(), batch)
            model.fit(data)
            t0 = time.perf_counter()
            for _ in range(samples):
                model.fit(data)
            out[f"{kind}_batch_sequences_{batch}_ms"] = 1e3 * (time.perf_counter() - t0) / samples
            if batch == 1:
                o = OracleModel(model.params.hp)
                t0 = time.perf_counter()
                o.fit(data.user_pointers, data.item_ids)
                out[f"{kind}_c_oracle_one_core_ms"] = 1e3 * (time.perf_counter() - t0)
                o.close()
    return out


def movielens_mrr():
    """BASELINE.json configs[1] (untimed): MovieLens-100K, LSTM Normal, dim 32, WARP, Adagrad,
    10 epochs under the reference protocol (lstm.rs:427-448, 498-520) — at batch_sequences 1, the
    reference's own schedule (one optimiser step per subsequence), and at 16.  `readme_example` times the
    configuration of the crate's README / doctest (lib.rs:22-58: max_sequence_length 32), whose fit the
    README describes as "about 10 seconds" (readme.md:26, unspecified CPU)."""
    from helpers import movielens_protocol
    from sbr_rs_amd._abi import make_hparams
    from sbr_rs_amd.engine import Model

    def run(max_len, batch):
        data, train, test, rng = movielens_protocol()
        hp = make_hparams(data.num_items(), max_len, 32, 0.16, 0.0004, 0, 2, 0, 1, rng.state_seed(), 10, 1, 0, batch)
        m = Model(hp)
        t0 = time.perf_counter()
        loss = m.fit(train.user_pointers, train.item_ids)
        dt = time.perf_counter() - t0
        mrr, ranks = m.mrr_score(test.user_pointers, test.item_ids)
        return {"test_mrr": mrr, "test_users": int(len(ranks)), "fit_loss": loss, "fit_seconds": dt,
                "train_interactions_per_s": 10 * (len(train.item_ids)) / dt}

    out = {"config": "MovieLens-100K, LSTM Normal, dim 32, WARP, Adagrad, 10 epochs, max_sequence_length 128, batch_sequences 16"}
    out.update(run(128, 16))
    out["batch_sequences_1"] = dict(run(128, 1), note="the reference's schedule: one optimiser step per subsequence (sequence_model.rs:111-169)")
    out["readme_example"] = dict(run(32, 1), config="lib.rs:22-58 / readme.md: max_sequence_length 32, dim 32, WARP, Adagrad, 10 epochs, batch_sequences 1",
                                 reference_says="about 10 seconds (readme.md:26, src/lib.rs:20; CPU unspecified)")
    return out


def mrr_gemm(model, args, users=8192):
    """`mrr_score` of the trained bench model over `users` synthetic test users against the whole catalogue (untimed
    extra): the rank kernels' time from the engine's HIP events, priced at 2 * users * items * dim flop."""
    tptr, titems = synthetic_csr(users, args.items, args.max_len, seed=7)
    model.mrr_score(tptr[:257], titems[: int(tptr[256])])  # warm-up (first launch, allocations)
    model.timing_enable(True)
    model.timing_read()
    t0 = time.perf_counter()
    mrr, ranks = model.mrr_score(tptr, titems)
    wall = time.perf_counter() - t0
    tm = model.timing_read()
    model.timing_enable(False)
    ms = tm["RANK"][0]
    flops = 2.0 * len(ranks) * args.items * args.dim
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "rank_test_score + rank_gemm (v_mfma_f32_32x32x2_f32, rank-count epilogue) + rank_history", "bound": "mfma",
            "users": int(len(ranks)), "items": args.items, "dim": args.dim, "rank_kernels_ms": ms, "launches": int(tm["RANK"][1]),
            "achieved": tf, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TF,
            "mrr_score_wall_ms": 1e3 * wall, "mrr": float(mrr), "note": "untimed extra; the U x I score matrix is never materialised"}


def quality_neutral_batch():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "quality_neutral_batch.json")))
    except Exception:
        return None


def workload_label(args, world):
    """Which BASELINE.json config the run's shape is (the label the JSON line carries)."""
    shape = (args.model, args.loss, args.dim, args.items, args.users, args.max_len)
    if shape == ("lstm", "warp", 128, 1_000_000, 100_000, 64) and world == 1 and not args.partition_table:
        return "BASELINE.json configs[2]"
    if shape == ("lstm", "warp", 128, 1_000_000, 125_000, 128) and world > 1 and not args.partition_table:
        return ("BASELINE.json configs[3]" if world == 8 else
                f"BASELINE.json configs[3]'s per-GPU shape at {world} GPUs (the config itself is 1M users over 8)")
    if shape == ("lstm", "warp", 128, 1_000_000, 125_000, 128) and world == 1 and not args.partition_table:
        return "BASELINE.json configs[3]'s per-GPU shape on ONE GPU (the same-shape denominator of the weak-scaling runs)"
    if (args.model, args.loss, args.dim, args.items) == ("ewma", "hinge", 256, 10_000_000) and args.partition_table:
        return ("BASELINE.json configs[4]" if world == 8 and args.users == 125_000 else
                "BASELINE.json configs[4]'s shape (EWMA + hinge, d 256, 1e7 items, partitioned item table)")
    return "custom workload"


def simulate_world(args, model_kind, loss_kind):
    """Rank 0's share of an N-GPU synchronous step on ONE GPU with ALL N ranks' real entries: N replicas (num_devices = N, rank q on
    its own partition) step together through the C-ABI halves, device-to-device copies standing in for the all-to-all / all-gather;
    the phases of RANK 0 are timed one by one (the other ranks' work is what fills rank 0's receive buffers with real touched-row
    densities — with N ranks nearly every table row is touched every step).  --exchange owner (default): scatter -> owner update of
    rank 0's slice (in place) -> parameter slices of the other owners copied into rank 0's table; --exchange gradient: rounds 1-5's
    owner reduce -> gathered gradient chunks -> whole-table update.  What it measures is the kernel-side term of the exchange; the
    link term is priced from the bytes.  (Round 5 fed rank 0's own chunk N times: one device's density.)"""
    import torch

    from sbr_rs_amd import engine
    from sbr_rs_amd._abi import Param
    from sbr_rs_amd.distributed import HipBackend, device_bytes_as_tensor

    n = args.simulate_world
    owner = args.exchange != "gradient"
    engine.set_device(0)
    ptr, items = synthetic_csr(args.users * n, args.items, args.max_len, zipf=args.item_distribution == "zipf")
    models = [engine.Model(make_hp(args, n, q, model_kind, loss_kind, args.items)) for q in range(n)]
    bes = [HipBackend(m, (ptr, items), n) for m in models]
    chunk, db = bes[0].chunk, bes[0].dense_bytes
    recv = [torch.zeros(n * chunk, dtype=torch.uint8, device="cuda") for _ in range(n)]
    dense_all = torch.zeros(n * db, dtype=torch.uint8, device="cuda")
    table = None if owner else torch.zeros(n * chunk, dtype=torch.uint8, device="cuda")
    blocks = {}
    for q in range(n):
        for which in (Param.ITEM_EMBEDDING, Param.ITEM_BIAS):
            p_, sb = models[q].table_slice(which)
            blocks[q, which] = (device_bytes_as_tensor(torch, p_, sb * n), sb)
    nmb = {be.epoch_prepare() for be in bes}
    assert len(nmb) == 1
    nmb = nmb.pop()
    names = ("local_compute", "scatter", "all_to_all_stand_in_copies", "owner_update" if owner else "owner_reduce", "all_gather_stand_in_copies") + \
            (() if owner else ("apply_rows",)) + ("dense_join_and_apply",)
    phases = {k: 0.0 for k in names}

    def timed(name, fn, on=True):
        if not on:
            fn()
            return
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        phases[name] += 1e3 * (time.perf_counter() - t0)

    rows = steps = 0
    for i in range(args.warmup + args.steps):
        mb = i % nmb
        if i == args.warmup:
            for k in phases:
                phases[k] = 0.0
            rows = steps = 0
        for q in range(n - 1, -1, -1):  # rank 0 last, its phases timed
            timed("local_compute", lambda: bes[q].compute_local(mb), q == 0)
            timed("scatter", lambda: bes[q].scatter(mb), q == 0)
        for p_ in range(n - 1, -1, -1):
            def a2a():  # chunk p of every device arrives at device p
                for src in range(n):
                    recv[p_][src * chunk:(src + 1) * chunk].copy_(bes[src].send[p_ * chunk:(p_ + 1) * chunk])
            timed("all_to_all_stand_in_copies", a2a, p_ == 0)
            if owner:
                timed("owner_update", lambda: bes[p_].plan.step_owner_update(recv[p_].data_ptr()), p_ == 0)
            else:
                timed("owner_reduce", lambda: bes[p_].owner_reduce(recv[p_]), p_ == 0)
        dn = [be.dense() for be in bes]
        for q in range(n - 1, -1, -1):
            def ag():
                if owner:  # the other owners' updated parameter slices, straight into this replica's table
                    for which in (Param.ITEM_EMBEDDING, Param.ITEM_BIAS):
                        dst, sb = blocks[q, which]
                        for p2 in range(n):
                            if p2 != q:
                                dst[p2 * sb:(p2 + 1) * sb].copy_(blocks[p2, which][0][p2 * sb:(p2 + 1) * sb])
                else:
                    for p2 in range(n):
                        table[p2 * chunk:(p2 + 1) * chunk].copy_(bes[p2].own)
            timed("all_gather_stand_in_copies", ag, q == 0)
            if not owner:
                timed("apply_rows", lambda: bes[q].apply_rows(table), q == 0)

            def dense_apply():
                for p2 in range(n):
                    dense_all[p2 * db:(p2 + 1) * db].copy_(dn[p2])
                bes[q].apply_dense(dense_all)
            timed("dense_join_and_apply", dense_apply, q == 0)
        rows += bes[0].plan.minibatch_rows(mb)
        steps += 1
    for be in bes:
        be.close()
    per = {k: v / steps for k, v in phases.items()}
    link_gbs = 153.0  # one xGMI link, per direction (the prompt's figure; 7 links per GPU, full mesh)
    kern = per["scatter"] + (per["owner_update"] if owner else per["owner_reduce"] + per["apply_rows"])
    return {
        "mode": f"simulate-world {n} on one GPU, all {n} ranks' real entries (kernel-side cost of the exchange; NOT a multi-GPU measurement)",
        "exchange": "owner-applied update, parameter slices gathered in place" if owner else "gradient all-gather, whole-table update on every replica (rounds 1-5)",
        "workload": f"{workload_label(args, n)}: rank 0 of {n}, {args.users} users/GPU x {args.items} items, seq_len<={args.max_len}, "
                    f"dim {args.dim}, {args.model}+{args.loss}, batch_sequences {args.batch_sequences}",
        "steps": steps, "interactions_per_step": rows / steps, "ms_per_phase": per,
        "exchange_kernels_ms": kern,
        "rank0_step_ms_without_links": per["local_compute"] + kern + per["dense_join_and_apply"],
        "single_device_update_ms_for_comparison": "see the N = 1 line's kernels.SPARSE_UPDATE",
        "chunk_bytes": chunk, "bytes_per_link_per_phase": chunk,
        "bytes_per_gpu_per_step": 2 * (n - 1) * chunk + (n - 1) * db,
        "link_time_estimate_ms": {"all_to_all": 1e3 * chunk / (link_gbs * 1e9), "all_gather": 1e3 * chunk / (link_gbs * 1e9),
                                  "assumption": f"every pair of GPUs exchanges one chunk per phase over its own link at {link_gbs} GB/s, all links concurrently"},
    }


def group_driver(args, model_kind, loss_kind):
    """`--driver group`: the path INTEGRATION.md binds for a one-process caller (Hyperparameters::num_threads(n) -> n device
    replicas -> sbr_group_fit): N replicas driven from THIS process through sbr_group_fit's own step sequence
    (sbr_group_fit_begin / _epoch_prepare / _step), replica r on HIP device r mod device count — N physical devices when the
    node has them, else N replicas sharing one GPU (then the devices' work is serialised: the host enqueue time is what that
    run measures, not scaling).  Timed with one host thread for all devices and with one host thread per device; `value` is the
    library's default mode.  ≙ /root/reference/src/models/sequence_model.rs:90-102 inside one process."""
    import zlib

    from sbr_rs_amd import engine
    from sbr_rs_amd._abi import Param

    n = args.gpus
    physical = engine.device_count()
    engine.set_device(0)
    ptr, items = synthetic_csr(args.users * n, args.items, args.max_len, zipf=args.item_distribution == "zipf")
    hp = make_hp(args, n, 0, model_kind, loss_kind, args.items)
    models = engine.group_create(hp, n, partition_item_table=args.partition_table)
    modes = {}

    def run(threads, gradient=args.exchange == "gradient"):
        gp = engine.GroupPlan(models, ptr, items, host_threads=threads)
        if gradient and n > 1 and not args.partition_table and args.parallelism != "async":
            gp.set_exchange(True)
        st = {"nmb": gp.epoch_prepare(prefetch_next=True), "mb": 0}

        def one():
            if st["mb"] >= st["nmb"]:
                st["nmb"], st["mb"] = gp.epoch_prepare(prefetch_next=True), 0
            mb = st["mb"]
            rows = sum(gp.member(r).minibatch_rows(mb) for r in range(n))
            gp.step(mb)
            st["mb"] += 1
            return rows

        for _ in range(args.warmup):
            one()
        gp.synchronize()
        q0, s0, nth = gp.stats()
        rows = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rows += one()
        t_queued = time.perf_counter()
        gp.synchronize()
        t1 = time.perf_counter()
        q1, s1, _ = gp.stats()
        gp.close()  # (gathers the owners' optimiser-state slices back onto every replica)
        return {"host_threads": nth, "exchange": ("partitioned table: owner-computes over gradient lists" if args.partition_table else "staleness-one pipeline (gradient all-gather)"
                                                  if args.parallelism == "async" else "gradient all-gather" if gradient else "owner-applied"), "ms_per_step": 1e3 * (t1 - t0) / args.steps, "host_enqueue_ms_per_step": (q1 - q0) / max(s1 - s0, 1),
                "host_loop_ms_per_step": 1e3 * (t_queued - t0) / args.steps, "interactions_per_s": rows / (t1 - t0), "interactions_timed": rows}

    default = run(None)
    modes["library_default"] = default
    if n > 1:
        modes["one_host_thread"] = run(False)
        modes["host_thread_per_device"] = run(True)
        if not args.partition_table and args.parallelism != "async":  # the other form of the Synchronous step, same box, same models (A/B)
            modes["library_default_other_exchange"] = run(None, gradient=args.exchange != "gradient")
    names = ["ITEM_EMBEDDING", "ITEM_EMBEDDING_ACC", "ITEM_BIAS", "ITEM_BIAS_ACC"] + (["LSTM_W", "LSTM_W_ACC", "LSTM_B"] if model_kind != 2 else ["EWMA_ALPHA"])
    crcs = []
    if args.param_crc:
        crcs = [{k: zlib.crc32(m.get_param(getattr(Param, k)).tobytes()) for k in names} for m in (models if not args.partition_table else models[:1] + models[-1:])]
    out = {
        "metric": "train interactions/sec", "value": default["interactions_per_s"], "unit": "interactions/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": default["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "driver": "group",
        "host_enqueue_ms_per_step": default["host_enqueue_ms_per_step"], "host_threads": default["host_threads"],
        "host_enqueue_fraction_of_step": default["host_enqueue_ms_per_step"] / default["ms_per_step"],
        "physical_devices": physical,
        "note": (f"{n} replicas on {physical} physical device(s)" + ("" if physical >= n else
                 ": the replicas SHARE a GPU, their work is serialised — this is a measurement of the group driver's host side "
                 "and of the exchange kernels, NOT of multi-GPU scaling")),
        "config": {"workload": f"{workload_label(args, n)}: {n} x {args.users} users x {args.items} items, seq_len<={args.max_len}, dim {args.dim}, "
                               f"{args.model}+{args.loss}, batch_sequences {args.batch_sequences} per replica, "
                               f"{'item table partitioned over the replicas (owner-computes)' if args.partition_table else 'table replicated (owner-reduce exchange)'}, "
                               f"{args.parallelism}", "parallelism": f"dp{n} in one process (sbr_group_fit)"},
        "modes": modes,
    }
    if args.partition_table:
        # remote-gather traffic of the partitioned table: (n-1)/n of the gathered rows live on a peer
        d = args.dim
        gather_bytes = (3 * 4 * d + 2 * 4)  # hinge / BPR: input, target, one negative + two biases; WARP: + (k-1) rows
        remote = gather_bytes * (n - 1) / n
        link = 153e9
        out["remote_gather"] = {
            "bytes_per_interaction_remote": remote, "bytes_per_interaction_gathered": gather_bytes,
            "xgmi_bound_interactions_per_s_per_gpu": (n - 1) * link / remote if n > 1 else None,
            "assumption": f"(n-1)/n of the gathered rows are remote; {n - 1} peers x 153 GB/s per direction, all links concurrently; "
                          "update lists travel the other way (3 reduced rows per interaction at most, before the per-row reduction)",
            "measured_here": "no (one physical device: every 'remote' row is local HBM)" if physical < n else "yes",
        }
    if crcs:
        out["param_crc_replicas"] = len(crcs)
        out["param_crc_replicas_equal"] = all(c == crcs[0] for c in crcs)
        out["param_crc"] = crcs[0]
    return out


def self_launch(n: int) -> int:
    """Re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on 127.0.0.1 with a free
    port (the user-sharded step needs one process per GPU, sequence_model.rs:90-102 ≙ DESIGN.md §8).  The ranks inherit stdout,
    so rank 0's single JSON line is this process's last line too; returns the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the peer mappings need it on this stack
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--users", type=int, default=None, help="users per GPU (default: 100 000 at N = 1 = configs[2]; 125 000 at N > 1 = configs[3])")
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--max-len", type=int, default=None, help="max_sequence_length (default: 64 at N = 1, 128 at N > 1)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch-sequences", type=int, default=None,
                    help="subsequences per optimiser step and GPU.  Default: the QUALITY-NEUTRAL batch — the largest one with unchanged "
                         "test MRR in the committed sweep (profiles/quality_neutral_batch.json: 8 192 for the LSTM; EWMA's MRR does not "
                         "fall up to 50 000, its default).  The 50 000-sequence figure (two optimiser steps per epoch of configs[2]: "
                         "the hardware's throughput regime, where the LSTM gives up a quarter of its MRR at equal epochs, NOTES.md §3) "
                         "is reported beside it as `value_max_batch`.")
    ap.add_argument("--item-distribution", choices=["uniform", "zipf"], default="uniform",
                    help="uniform = the pure-roofline run (no cache reuse); zipf = Zipf(1.0) over a permuted catalogue")
    ap.add_argument("--model", choices=["lstm", "lstm-coupled", "ewma"], default="lstm")
    ap.add_argument("--loss", choices=["bpr", "hinge", "warp"], default="warp")
    ap.add_argument("--parallelism", choices=["sync", "async"], default="sync",
                    help="multi-GPU step: sync = Parallelism::Synchronous (the reference default); async = the "
                         "staleness-one pipeline (Parallelism::Asynchronous): compute k+1 under the exchange of step k")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="worker threads of the CPU baseline (0 = every host core, the reference's default num_threads, lstm.rs:68); "
                         "all of them share ONE model")
    ap.add_argument("--cpu-users", type=int, default=100_000, help="users in the CPU-baseline sample (capped at --users)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="wall-time bound of each CPU-baseline leg (1, 16, 64 and all threads)")
    ap.add_argument("--standalone-steps", type=int, default=6,
                    help="extra untimed steps with stream overlap disabled, for standalone per-kernel times (0 = skip)")
    ap.add_argument("--cold-items", type=int, default=4_000_000,
                    help="catalogue size of the untimed cache-cold pass of the gather + score kernel (table = 8x the 256 MiB "
                         "Infinity Cache at dim 128); 0 = skip")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="one GPU: rank 0's share of an N-GPU synchronous step with the real exchange kernels (scatter into N chunks, "
                         "owner reduce over N inputs, table update) and device copies in place of the collectives; prints kernel ms "
                         "and bytes per link instead of the throughput line's usual extras")
    ap.add_argument("--batch-sweep", type=str, default="1024,4096,8192,16384",
                    help="extra batch sizes measured untimed after the main run (reported with the main one as batch_sweep); '' = skip")
    ap.add_argument("--traffic", choices=["live", "profile", "off"], default="live",
                    help="roofline.traffic (HBM bytes per launch of the gather + score kernel): live = measured in this very invocation by "
                         "two rocprofv3 PMC passes around a child run at the same operating point (N = 1; falls back to `profile` if "
                         "rocprofv3 is unavailable); profile = the committed profile of the same operating point "
                         "(profiles/score_kernel_traffic.json) or null; off = null")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mrr", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend; gloo (host-staged) lets several ranks share one GPU for testing the N > 1 path")
    ap.add_argument("--transport", choices=["collective", "peer"], default="collective",
                    help="replicated multi-GPU exchange: collective = RCCL all-to-all + all-gather of the chunks; peer = the "
                         "kernels read the peers' chunk buffers in place through peer mappings (xGMI), collectives only order the phases")
    ap.add_argument("--partition-table", action="store_true",
                    help="store the item table once across the ranks (BASELINE configs[4] layout) instead of replicating it")
    ap.add_argument("--param-crc", action="store_true",
                    help="add CRC-32 checksums of the trained parameters to the JSON line (tests compare runs bit for bit)")
    ap.add_argument("--driver", choices=["ranks", "group"], default="ranks",
                    help="ranks: one process per GPU over torch.distributed (the launcher contract); group: --gpus N replicas driven "
                         "from ONE process through sbr_group_fit's step sequence (what INTEGRATION.md's Rust binding calls)")
    ap.add_argument("--exchange", choices=["owner", "gradient"], default="owner",
                    help="Synchronous multi-GPU step over a replicated table: owner = the owner of a slice reduces AND updates it in place, "
                         "the updated parameter slices are all-gathered into every replica's table (round 6); gradient = rounds 1-5: the "
                         "reduced gradient chunks are all-gathered and every replica updates the whole table.  Same bits")
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="keep the GPU busy with a model-neutral load for this long before the warm-up steps (see the comment at its use)")
    ap.add_argument("--scale-shape", action="store_true",
                    help="N = 1 on the shape the N > 1 runs use per GPU (BASELINE configs[3]: 125 000 users, seq_len <= 128): the denominator of a "
                         "weak-scaling efficiency (every N > 1 line also carries it as `single_gpu_same_shape`)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU exchange collectives even at world size 1 (smoke test of the RCCL path)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.simulate_world <= 1 and args.driver != "group":
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's launcher would
        # (python -m torch.distributed.run, one rank per GPU); rank 0's JSON line stays the last line of stdout
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus and args.driver != "group":
        args.gpus = world
    multi = world > 1 or args.simulate_world > 1 or (args.driver == "group" and args.gpus > 1) or args.scale_shape
    if args.users is None:
        args.users = 125_000 if multi else 100_000   # BASELINE.json configs[3] (1M users over 8 GPUs) / configs[2]
    if args.max_len is None:
        args.max_len = 128 if multi else 64
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    shared_devices_note = None
    if int(os.environ.get("LOCAL_WORLD_SIZE", world)) > torch.cuda.device_count() and args.backend != "gloo":
        # more ranks than devices on this node (a one-GPU box): RCCL refuses two ranks on one device; the host-staged test transport
        # does not.  Every rank takes this branch (LOCAL_WORLD_SIZE is the launcher's), and the line says what it is.
        shared_devices_note = (f"{os.environ.get('LOCAL_WORLD_SIZE', world)} ranks on {torch.cuda.device_count()} device(s): ranks share a GPU, "
                               "gloo (host-staged) transport — a functional run, not a scaling measurement")
        args.backend = "gloo"
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # ranks may share a device
    torch.cuda.set_device(local_rank)
    from sbr_rs_amd import engine
    from sbr_rs_amd._abi import Debug
    from sbr_rs_amd.distributed import HipBackend, StepLoop

    engine.set_device(local_rank)
    dist = None
    backend_note = shared_devices_note
    if world > 1 or args.force_exchange:
        import torch.distributed as dist

        if world == 1:  # --force-exchange without a launcher
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")):
                os.environ.setdefault(k, v)
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            try:  # RCCL over xGMI; a first collective here so that a transport problem shows before the plan is built
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as e:  # no usable RCCL on this host: the host-staged test transport still produces a (slower) line
                backend_note = f"nccl unavailable ({e!r}): gloo, host-staged"
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo")
                args.backend = "gloo"

    model_kind = {"lstm": 0, "lstm-coupled": 1, "ewma": 2}[args.model]
    loss_kind = {"bpr": 0, "hinge": 1, "warp": 2}[args.loss]
    MAX_BATCH = 50_000
    batch_rule = "--batch-sequences"
    if args.batch_sequences is None:
        qn = quality_neutral_batch()
        if model_kind != 2 and qn:
            args.batch_sequences = int(qn["batch_sequences"])
            batch_rule = f"default: the quality-neutral batch ({qn['table']}: {qn['criterion']})"
        else:
            args.batch_sequences = MAX_BATCH
            batch_rule = "default: 50 000 (EWMA: test MRR does not fall with the batch, NOTES.md section 3)" if model_kind == 2 else "default: 50 000"
    if args.driver == "group":
        if world != 1:
            raise SystemExit("--driver group runs in one process (no launcher)")
        print(json.dumps(group_driver(args, model_kind, loss_kind)), flush=True)
        return
    if args.simulate_world > 1:
        if world != 1:
            raise SystemExit("--simulate-world runs in one process on one GPU")
        print(json.dumps(simulate_world(args, model_kind, loss_kind)), flush=True)
        return
    total_users = args.users * world
    ptr, items = synthetic_csr(total_users, args.items, args.max_len, zipf=args.item_distribution == "zipf")
    hp = make_hp(args, world, rank, model_kind, loss_kind, args.items)
    if args.partition_table and world > 1:
        # the item table stored once across the ranks; torch.distributed is the control plane only
        from sbr_rs_amd.partitioned import PartitionedStepper, create_partitioned_model

        model = create_partitioned_model(hp)
        backend = loop = PartitionedStepper(model, (ptr, items))
        plan = backend.plan
    elif args.transport == "peer" and (world > 1 or args.force_exchange):
        from sbr_rs_amd.partitioned import PeerExchangeStepper

        model = engine.Model(hp)
        backend = loop = PeerExchangeStepper(model, (ptr, items))
        plan = backend.plan
    else:
        model = engine.group_create(hp, 1, partition_item_table=True)[0] if args.partition_table else engine.Model(hp)
        backend = HipBackend(model, (ptr, items), world if not args.force_exchange else max(world, 2))
        plan = backend.plan
        if args.force_exchange and world == 1:  # one rank owns the whole table: a single chunk
            backend.send = backend.send[:backend.chunk]
        # the production step sequencing (sbr_rs_amd/distributed.py); --force-exchange runs the collectives at world 1
        loop = StepLoop(backend, world if not args.force_exchange else max(world, 2), asynchronous=args.parallelism == "async",
                        exchange=args.exchange)
        if args.force_exchange and world == 1:
            loop.world = 2  # take the exchange branch; the process group itself has a single rank
            backend.world = 1  # (one owner slice: the whole table)
            c, db = backend.chunk, backend.dense_bytes
            loop.bufs = tuple(torch.zeros(n, dtype=torch.uint8, device="cuda") for n in (c, c, db))

    tp0 = time.perf_counter()
    state = {"nmb": loop.begin_epoch(prefetch_next=True), "mb": 0, "reprepared_in_timed_region": 0, "epoch_switch_ms": 0.0}
    epoch_prepare_ms = 1e3 * (time.perf_counter() - tp0)

    def one_step(timed: bool) -> int:
        if state["mb"] >= state["nmb"]:
            t_e = time.perf_counter()
            state["nmb"] = loop.begin_epoch(prefetch_next=True)  # epoch switch; the host packed it in the background
            state["mb"] = 0
            if timed:
                state["reprepared_in_timed_region"] += 1
                state["epoch_switch_ms"] += 1e3 * (time.perf_counter() - t_e)  # host time of the switch: waits for the packing thread if it is behind
        mb = state["mb"]
        rows = plan.minibatch_rows(mb)
        loop.step(mb)  # compute_local + (exchange: scatter, all-to-all, owner reduce, all-gather) + apply
        state["mb"] += 1
        return rows

    def sync():
        model.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    prewarm = None
    if args.prewarm_seconds > 0 and torch.cuda.is_available():
        # The timed region is K steps of a few milliseconds each, entered seconds after the process has generated its synthetic data on
        # the host with the GPU idle: without this the K steps are measured on a GPU that is still leaving its idle power state (the
        # HBM-bound kernels read ~10 % slow, profiles/r06_prewarm.md).  A model-neutral load (no optimiser step, no model state
        # touched: a torch f32 matmul and a device copy) keeps the GPU busy for --prewarm-seconds first; the W warm-up steps and the
        # K timed steps follow immediately, unchanged.
        t_w = time.perf_counter()
        wa = torch.empty(4096, 4096, device="cuda").normal_()
        wb = torch.empty(4096, 4096, device="cuda").normal_()
        wc = torch.empty(4096, 4096, device="cuda")
        big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        big2 = torch.empty_like(big)
        n_w = 0
        while time.perf_counter() - t_w < args.prewarm_seconds:
            for _ in range(4):
                torch.mm(wa, wb, out=wc)
                big2.copy_(big)
            torch.cuda.synchronize()
            n_w += 4
        prewarm = {"seconds": time.perf_counter() - t_w, "what": "torch f32 4096^3 matmul + 256 MiB device copy in a loop before the warm-up steps; touches no model state",
                   "iterations": n_w}
        del wa, wb, wc, big, big2
    for _ in range(args.warmup):
        one_step(False)
    sync()
    # SBR_BENCH_TIMERS = all | score | off: which kernel families are bracketed by HIP events inside the timed region (A/B of the
    # events' own cost; the roofline needs the score family's)
    # Inside the timed region only the roofline's kernel family is bracketed by HIP events: two event records per bracketed launch
    # cost a 2.5 ms step of ~20 launches 1.5-2 % (profiles/r04_tail_experiments.md).  The per-family times of `kernels` /
    # `roofline_mfma` come from a second pass of the same number of steps, same schedule, with every family bracketed — outside the
    # timed region.  SBR_BENCH_TIMERS = all | score | off is the A/B switch.
    timers = os.environ.get("SBR_BENCH_TIMERS", "score")
    if hasattr(model, "timing_select"):
        model.timing_select(None if timers == "all" else ["SCORE"] if timers == "score" else [])
    model.timing_enable(True)
    model.timing_read()  # reset
    ex0, neg0 = plan.counters()
    rows_timed = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows_timed += one_step(True)
    model.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        rdev = "cpu" if args.backend == "gloo" else "cuda"
        per_rank = [None] * world
        dist.all_gather_object(per_rank, 1e3 * elapsed / max(args.steps, 1))
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rows_timed], dtype=torch.int64, device=rdev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rows_total = int(r.item())
        dist.barrier()
    else:
        rows_total = rows_timed
        per_rank = [1e3 * elapsed / max(args.steps, 1)]
    timing = model.timing_read()
    ex1, neg1 = plan.counters()
    score_timing = timing.get("SCORE")
    rows_families = rows_timed
    if timers != "all" and hasattr(model, "timing_select"):
        # second pass, untimed for `value`: the same steps with every kernel family bracketed by events
        model.timing_select(None)
        rows_families = 0
        for _ in range(args.steps):
            rows_families += one_step(False)
        sync()
        timing = model.timing_read()
        if score_timing and score_timing[1]:
            timing["SCORE"] = score_timing  # the roofline kernel's time is the timed region's
    sparse_entries, sparse_unique = plan.sparse_stats() if hasattr(plan, "sparse_stats") else (0, 0)
    # Second, UNTIMED pass (single GPU): the same steps with the side-stream work queued on the main stream,
    # so that every kernel family runs alone — the times behind the per-kernel roofline figures below.  The
    # timed region above keeps the overlapped schedule and is what `value` reports.
    standalone, rows_standalone = None, 0
    if world == 1 and not args.force_exchange and args.standalone_steps > 0:
        model.set_overlap(False)
        model.timing_read()
        for _ in range(args.standalone_steps):
            rows_standalone += one_step(False)
        sync()
        standalone = model.timing_read()
        model.set_overlap(True)
    model.timing_enable(False)

    def short_run(hp_x, ptr_x, items_x, steps, warm, timers=True):
        """A few untimed-region steps of another configuration on a fresh model: (interactions/s, SCORE ms per launch,
        rows per launch, negatives per interaction)."""
        mdl = engine.Model(hp_x)
        be = HipBackend(mdl, (ptr_x, items_x), 1)
        lp = StepLoop(be, 1, asynchronous=False)
        nmb = lp.begin_epoch(prefetch_next=False)
        for i in range(warm):
            lp.step(i % nmb)
        mdl.synchronize()
        if hasattr(mdl, "timing_select"):
            mdl.timing_select(["SCORE"])  # only the roofline kernel's launches are bracketed (as in the main timed region)
        mdl.timing_enable(timers)  # (the per-kernel events cost a small step a third of its time: off for those)
        mdl.timing_read()
        e0, n0 = be.plan.counters()
        rows = 0
        t_a = time.perf_counter()
        for i in range(steps):
            mbi = (warm + i) % nmb
            rows += be.plan.minibatch_rows(mbi)
            lp.step(mbi)
        mdl.synchronize()
        dt = time.perf_counter() - t_a
        tm = mdl.timing_read()
        e1, n1 = be.plan.counters()
        be.close()
        sc_ms = tm["SCORE"][0] / max(tm["SCORE"][1], 1) if timers and "SCORE" in tm else None
        return rows / dt, sc_ms, rows / max(steps, 1), (n1 - n0) / max(e1 - e0, 1), 1e3 * dt / max(steps, 1)

    cold, sweep, small = None, None, None
    if world == 1 and not args.force_exchange and not args.partition_table and rank == 0:
        if args.cold_items > 0 and model_kind != 2:
            try:  # the same kernel against a table far larger than the Infinity Cache: no row is re-read from cache
                cptr, citems = synthetic_csr(args.users, args.cold_items, args.max_len)
                _, sc_ms, rpl, kc, _ = short_run(make_hp(args, 1, 0, model_kind, loss_kind, args.cold_items), cptr, citems, 3, 1)
                cb = ((2 + kc) * 4 * args.dim + (1 + kc) * 4) * rpl
                ct, ctl = measured_traffic("cold", rpl, kc, args.dim, args.cold_items)
                cold = {"items": args.cold_items, "table_bytes": args.cold_items * args.dim * 4, "avg_launch_ms": sc_ms,
                        "mean_negatives_scored": kc, "algorithmic_bytes_per_launch": cb, "traffic": ct, "traffic_lower": ctl,
                        "achieved": cb / (sc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": cb / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                del cptr, citems
            except Exception as e:
                cold = {"error": repr(e)}
        if args.batch_sweep:
            sweep = []
            try:
                qn = quality_neutral_batch()
                sizes = sorted({int(x) for x in args.batch_sweep.split(",") if x} | ({int(qn["batch_sequences"])} if qn else set()) | {MAX_BATCH})
                for bsz in [b for b in sizes if b != args.batch_sequences]:
                    big = bsz == MAX_BATCH  # the max-batch figure is a reported value: the driver's step counts, not the sweep's short ones
                    v, sc_ms, rpl, kk, ms = short_run(make_hp(args, 1, 0, model_kind, loss_kind, args.items, batch=bsz), ptr, items,
                                                      args.steps if big else 8, args.warmup if big else 3)
                    sweep.append({"batch_sequences": bsz, "interactions_per_s": v, "ms_per_step": ms, "interactions_per_step": rpl})
                    if big and sc_ms:
                        bb = ((2 + kk) * 4 * args.dim + (1 + kk) * 4) * rpl
                        sweep[-1]["score_kernel"] = {"avg_launch_ms": sc_ms, "mean_negatives_scored": kk, "algorithmic_bytes_per_launch": bb,
                                                     "achieved": bb / (sc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": bb / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            except Exception as e:
                sweep.append({"error": repr(e)})
        if args.batch_sweep and workload_label(args, 1).startswith("BASELINE.json configs[2]"):
            # the reference's own regime: embedding_dim 32 (lib.rs:22-58) and few sequences per optimiser step — wave-per-sequence
            # recurrent kernels, block-form dense gradient, single-launch ordering / reduction (profiles/r03_small_steps.md)
            small = {"config": "synthetic 40 000 users x 100 000 items, len <= 64, dim 32, lstm+warp, whole optimiser steps, no per-kernel timers",
                     "rows": []}
            try:
                sp, si = synthetic_csr(40_000, 100_000, 64)
                for bsz in (1, 16, 256, 2048):
                    v, _, rpl, _, ms = short_run(make_hp(args, 1, 0, 0, 2, 100_000, batch=bsz, dim=32, max_len=64), sp, si, 200 if bsz <= 256 else 19, 40 if bsz <= 256 else 1, timers=False)
                    small["rows"].append({"batch_sequences": bsz, "interactions_per_s": v, "ms_per_step": ms, "interactions_per_step": rpl})
                del sp, si
            except Exception as e:
                small["error"] = repr(e)

    if hasattr(loop, "finish"):
        loop.finish()  # owner-applied steps: the owners' optimiser-state slices back on every replica (outside every timed region)
    single_same_shape = None
    if world > 1 and not args.partition_table:
        # the denominator of the weak-scaling efficiency, measured in THIS job: rank 0 alone on the same per-GPU shape and batch (the
        # N = 1 default of this script is configs[2]'s shorter sequences, a different workload); the other ranks wait at the barrier
        if rank == 0:
            try:
                sptr, sitems = synthetic_csr(args.users, args.items, args.max_len, zipf=args.item_distribution == "zipf")
                v1, _, rpl1, _, ms1 = short_run(make_hp(args, 1, 0, model_kind, loss_kind, args.items), sptr, sitems, args.steps, args.warmup, timers=False)
                single_same_shape = {"value": v1, "unit": "interactions/s", "ms_per_step": ms1, "interactions_per_step": rpl1, "steps": args.steps,
                                     "warmup": args.warmup, "what": "rank 0's GPU alone, one process, no exchange: the same users per GPU, sequence "
                                                                  "lengths, model and batch_sequences as every rank of this run"}
                del sptr, sitems
            except Exception as e:
                single_same_shape = {"error": repr(e)}
        dist.barrier()
    crc_ranks = None
    if args.param_crc:  # every rank's replica (a partitioned table is read whole through the rank's mapping)
        import zlib

        from sbr_rs_amd._abi import Param

        names = ["ITEM_EMBEDDING", "ITEM_EMBEDDING_ACC", "ITEM_BIAS", "ITEM_BIAS_ACC"] + (["LSTM_W", "LSTM_W_ACC", "LSTM_B"] if model_kind != 2 else ["EWMA_ALPHA"])
        own = {n: zlib.crc32(model.get_param(getattr(Param, n)).tobytes()) for n in names}
        crc_ranks = [own]
        if dist is not None and world > 1:
            crc_ranks = [None] * world
            dist.all_gather_object(crc_ranks, own)
    if rank == 0:
        d, ng = args.dim, {0: 4, 1: 3, 2: 0}[model_kind]
        # negatives actually scored per interaction over the timed steps (all devices)
        k_mean = (neg1 - neg0) / max(ex1 - ex0, 1)
        rows_per_launch = rows_timed / max(args.steps, 1)
        kernels = {}
        for name, (ms, n) in timing.items():
            if n:
                kernels[name] = {"ms_total": ms, "launches": int(n), "ms_per_launch": ms / n}
        if "SPARSE_SORT" in kernels and world == 1 and loss_kind == 2:
            kernels["SPARSE_SORT"]["note"] = ("elapsed on its own stream from the end of the score kernel, underneath the backward pass whose MFMA waves it "
                                              "yields to (wave priority 1; placement and priority A/B: profiles/r04_tail_experiments.md); alone: "
                                              "kernels_standalone; the step's loss bookkeeping (seq_loss, block_header) rides on the same stream")
        # roofline of the gather + WARP-score kernel: algorithmic bytes per packed row:
        # (2+k)*4d for h, the positive row and k negative rows, (1+k)*4 for their biases
        score = kernels.get("SCORE")
        roofline = None
        if score:
            bytes_per_row = (2 + k_mean) * 4 * d + (1 + k_mean) * 4
            # EWMA + single-negative loss since round 6: the sequence's backward scan runs in the same launch (no RECURRENT_BWD family);
            # the launch is then priced at the scan + score bytes plus the BPTT intermediates no decomposition avoids — h written,
            # s_{t-1} read, dX written (3 x 4d) — and NOT at the backward scan's own gather of E[in], E[neg] (8d: `step_bytes.two_pass`)
            ewma_whole = model_kind == 2 and loss_kind != 2 and "RECURRENT_BWD" not in kernels
            if ewma_whole:
                bytes_per_row += 12 * d
            bytes_per_launch = bytes_per_row * rows_per_launch
            achieved = bytes_per_launch / (score["ms_per_launch"] * 1e-3) / 1e9
            # HBM bytes per launch MEASURED for this very configuration: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
            # runs) around the default bench command, the timed dispatches only, corrected per profiles/r03_counter_calibration.md;
            # printed only when this run's rows per launch and mean k are within 5 % of the profiled run's, else null
            traffic = traffic_lower = None
            traffic_source = traffic_step = None
            pmc_kernel = "ewma_seq_kernel" if model_kind == 2 and loss_kind != 2 else "score_kernel" if loss_kind == 2 else "score_single_kernel"
            if args.traffic == "live" and world == 1 and not args.force_exchange:
                lt = live_traffic(pmc_kernel, args.warmup, args.steps)
                # the child's operating point must be this run's (same command line): rows per launch and negatives per row within 5 %
                if lt and abs(lt["rows_per_launch"] / max(rows_per_launch, 1) - 1) <= 0.05 and abs(lt["mean_negatives_scored"] / max(k_mean, 1e-9) - 1) <= 0.05:
                    traffic, traffic_lower = lt["hbm_bytes_per_launch"], lt["hbm_bytes_per_launch_lower"]
                    traffic_step = lt.get("step")
                    traffic_source = (f"measured in this invocation: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) around a child run of "
                                      f"the same command line, the {lt['dispatches_averaged']} timed dispatches of {pmc_kernel}; FETCH_SIZE {lt['FETCH_SIZE_KiB_mean']:.0f} KiB, "
                                      f"WRITE_SIZE {lt['WRITE_SIZE_KiB_mean']:.0f} KiB per launch; bytes = (2 F + W) x 1024, lower figure = minus 64 B per bias read "
                                      "(profiles/r03_counter_calibration.md)")
            if traffic is None and args.traffic != "off":
                traffic, traffic_lower = measured_traffic("warm", rows_per_launch, k_mean, d, args.items)
                traffic_source = "profiles/score_kernel_traffic.json (PMC passes of the same command at this operating point; interval: profiles/r03_counter_calibration.md)" if traffic else None
            kname = ("ewma_seq_kernel<D> (EWMA scan + gather + negative + loss + backward scan of a sequence in one pass; priced at "
                     "(2+k)4d + (1+k)4 + 12d per row: section 8d's gather + score bytes, h written, s_{t-1} read, dX written; sbr_kernels.hip)" if ewma_whole else
                     "ewma_seq_kernel (EWMA scan + gather + negative + loss in one pass per sequence; x_t replaces the h_t read, h_t is written "
                     "once on top of the priced bytes; sbr_kernels.hip)" if model_kind == 2 and loss_kind != 2 else
                     "score_kernel (gather + negative sampling + loss, sbr_kernels.hip)" if loss_kind == 2 else
                     "score_single_kernel (gather + negative + loss, sbr_kernels.hip)")
            roofline = {"kernel": kname, "bound": "hbm",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_lower": traffic_lower, "traffic_step": traffic_step,
                        "traffic_source": traffic_source, "traffic_over_algorithmic": [traffic_lower / bytes_per_launch, traffic / bytes_per_launch] if traffic else None,
                        "algorithmic_bytes_per_launch": bytes_per_launch,
                        "rows_per_launch": rows_per_launch, "mean_negatives_scored": k_mean,
                        "avg_launch_ms": score["ms_per_launch"],
                        "cache_policy": ("achieved = algorithmic bytes / time against the HBM peak; `traffic` counts what the L2s fetch (FETCH_SIZE), i.e. HBM plus "
                                         "Infinity Cache.  Since round 5 the forward pass streams everything but h past the caches, so the h rows' share of the "
                                         "algorithmic bytes (4d of (2+k)4d per row) is read from the Infinity Cache; table rows and h rows are read nt "
                                         "(DESIGN.md section 6, profiles/r05_streaming_gathers.md)") if loss_kind == 2 and model_kind != 2 else None}
            ceil = measured_ceiling(4 * d, cached_table=args.items * d * 4 < (1 << 30))
            if ceil and traffic:
                roofline["measured_gather_ceiling"] = {"GBps": ceil, "source": "tools/hbm_ceiling.hip, profiles/r02_hbm_ceiling.jsonl",
                                                       "real_traffic_GBps": traffic / (score["ms_per_launch"] * 1e-3) / 1e9,
                                                       "real_traffic_frac_of_ceiling": traffic / (score["ms_per_launch"] * 1e-3) / 1e9 / ceil}
        # standalone kernel times (second pass): HBM figure of the sparse update (3 rows x
        # (gradient source 4d, w and G read, w and G written) + biases = 36d + 24 B per interaction) and MFMA
        # figures of the three GEMM-shaped kernels without their stream partners
        kernels_sa = None
        if standalone:
            per_row = rows_standalone and {n: ms / rows_standalone for n, (ms, c) in standalone.items() if c}
            kernels_sa = {"steps": args.standalone_steps, "interactions": rows_standalone,
                          "ms_per_step": {n: ms / args.standalone_steps for n, (ms, c) in standalone.items() if c}}
            if "SPARSE_UPDATE" in per_row:
                # (a) BASELINE.md §4's per-interaction formula: 3 rows x (gradient source 4d, w and G read, w and G written)
                #     + biases = 36d + 24 B — what a per-entry update would move;
                # (b) the bytes this update really moves: it reduces the entries per table row first, so every DISTINCT row
                #     is read-modified-written once (w, G in and out: 16d B, + 16 B of bias state for rows that have one)
                #     and every entry reads its 4d-byte gradient source row (+ 4 B coefficient, 8 B sorted key)
                sa_step_ms = kernels_sa["ms_per_step"]["SPARSE_UPDATE"]
                gbs = (36 * d + 24) / (per_row["SPARSE_UPDATE"] * 1e-3) / 1e9
                real_bytes = sparse_unique * (16 * d + 16) + sparse_entries * (4 * d + 12)
                real_gbs = real_bytes / (sa_step_ms * 1e-3) / 1e9 if sparse_entries else None
                kernels_sa["sparse_update_hbm"] = {
                    "kernel": "seg_short_kernel (+ hot-row path): per-row reduction of the sorted entries + one Adagrad read-modify-write per distinct row",
                    "bound": "hbm", "achieved": real_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": real_gbs / HBM_PEAK_GBS if real_gbs else None,
                    "bytes_per_launch": real_bytes, "entries_per_launch": sparse_entries, "distinct_rows_per_launch": sparse_unique,
                    "pricing": "distinct rows x (16d + 16) + entries x (4d + 12) bytes (de-duplicated, what the kernel moves)",
                    "per_entry_formula": {"bytes_per_interaction": 36 * d + 24, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                                          "note": "BASELINE.md section 4 formula (no de-duplication): an upper bound on the traffic, not what moves"}}
            if ng:
                gemm_sa = 2 * 2 * d * ng * d
                kernels_sa["mfma"] = {fam: {"achieved": gemm_sa / (per_row[fam] * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                            "frac": gemm_sa / (per_row[fam] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF}
                                      for fam in ("RECURRENT_FWD", "RECURRENT_BWD", "DENSE_GRAD") if fam in per_row}
        mfma = []
        if ng:
            gemm = 2 * 2 * d * ng * d  # flop per packed row of one [x;h] x W sized GEMM
            for fam, flops_per_row, what in (("RECURRENT_FWD", gemm, "sequence-resident LSTM forward (gate GEMM + cell)"),
                                             ("RECURRENT_BWD", gemm, "sequence-resident BPTT (cell backward + dz W^T GEMM)"),
                                             ("DENSE_GRAD", gemm, "dense-gradient GEMM xh^T dz (runs on the side stream, overlapping the sparse update)")):
                if fam in kernels:
                    tf = flops_per_row * rows_families / (kernels[fam]["ms_total"] * 1e-3) / 1e12
                    mfma.append({"kernel": fam, "what": what, "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TF,
                                 "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TF})
        # EWMA + single-negative loss (configs[4]'s step): a purely memory-bound step — its byte budget per optimiser step
        # (profiles/r06_ewma_bytes.md) beside what the PMC counters saw over the same steps
        step_bytes = None
        if model_kind == 2 and loss_kind != 2 and world == 1 and sparse_entries and not args.partition_table:
            R_, B_ = rows_per_launch, min(args.batch_sequences, args.users)
            fwd = R_ * (2 * 4 * d + 4 * d + 8 + 12 + 24) + B_ * 4 * d          # target + negative rows, H written, biases, ids, result words; step 0's input
            bwd = R_ * (4 * d + 4 * d + 12) + B_ * 4 * d                        # s_{t-1} read, dX written, ids / coefficient; the last target row
            regather = R_ * 2 * 4 * d                                           # the backward scan's own gather of E[in], E[neg] (forward had them in registers)
            upd = sparse_unique * (16 * d + 16) + sparse_entries * (4 * d + 12)  # distinct rows read-modify-written once, one source row per entry
            srt = sparse_entries * 8 * 2 * 3                                    # three radix passes over 64-bit keys
            step_s = elapsed / max(args.steps, 1)
            necessary, two_pass, formula = fwd + bwd + upd + srt, fwd + bwd + regather + upd + srt, (12 * d + 8 + 36 * d + 24) * R_
            step_bytes = {"rows_per_step": R_, "entries_per_step": sparse_entries, "distinct_rows_per_step": sparse_unique,
                          "formula_8d": formula, "necessary": necessary, "two_pass": two_pass,
                          "parts": {"scan_score": fwd, "backward_scan": bwd, "backward_regather": regather, "update": upd, "sort": srt},
                          "frac_of_hbm_peak": {"formula_8d": formula / step_s / 1e9 / HBM_PEAK_GBS, "necessary": necessary / step_s / 1e9 / HBM_PEAK_GBS,
                                               "two_pass": two_pass / step_s / 1e9 / HBM_PEAK_GBS},
                          "what": ("necessary = scan + score (target and negative rows gathered, h written once) + backward scan (s_{t-1} read, dX written) + "
                                   "update (distinct rows x (16d + 16) read-modify-written once + entries x (4d + 12) of gradient source and key) + three radix "
                                   "passes; two_pass = necessary + the backward scan's own gather of E[in], E[neg] (8d per row: the rows the forward scan held in "
                                   "registers; a sequence's rows do not stay on chip between the two scans at this size); formula_8d = SURVEY section 8d's "
                                   "(2+k)4d + (1+k)4 + 36d + 24 per row, which prices no BPTT intermediate and no de-duplication")}
            if roofline and roofline.get("traffic_step"):
                ts = roofline["traffic_step"]
                step_bytes["measured"] = ts["hbm_bytes_per_step"]
                step_bytes["measured_over_necessary"] = ts["hbm_bytes_per_step"] / necessary
                step_bytes["measured_rate_GBps"] = ts["hbm_bytes_per_step"] / step_s / 1e9
                step_bytes["measured_rate_note"] = ("read + write traffic of every kernel of the step over the step's wall time; this device's ceilings for such a mix "
                                                    "(tools/hbm_ceiling.hip, profiles/r02_hbm_ceiling.jsonl): copy 4.6 TB/s, random 1 KiB row read-modify-write 4.95, read-only gather 6.0-6.2")
        workload_tag = workload_label(args, world)
        out = {
            "metric": "train interactions/sec", "value": rows_total / elapsed, "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload_tag}: synthetic {args.users} users/GPU x {args.items} items, "
                                   f"seq_len<={args.max_len}, dim {args.dim}, {args.model}+{args.loss}, Adagrad lr 0.16 l2 4e-4",
                       "users_per_gpu": args.users, "items": args.items, "max_len": args.max_len, "dim": args.dim,
                       "batch_sequences_per_gpu": args.batch_sequences, "batch_rule": batch_rule, "item_distribution": args.item_distribution,
                       "item_table": "partitioned across the ranks (one copy)" if args.partition_table else "replicated",
                       "parallelism": (f"user-sharded dp{world}, {'staleness-one pipelined (Asynchronous)' if args.parallelism == 'async' else 'synchronous'} "
                                       f"owner-reduce exchange over {'RCCL' if args.backend == 'nccl' else 'gloo (host-staged, test transport)'}") if world > 1 else "single device"},
            "process_group_ranks": dist.get_world_size() if dist is not None else 1,
            "rccl_ranks": dist.get_world_size() if dist is not None and args.backend == "nccl" else 0,
            "collective_backend": (backend_note or (args.backend if args.backend == "gloo" else "nccl (RCCL)")) if dist is not None else None,
            "ms_per_step_per_rank": per_rank,
            "interactions_timed": rows_total, "epoch_prepares_in_timed_region": state["reprepared_in_timed_region"],
            "epoch_prepare_ms": epoch_prepare_ms, "epoch_switch_host_ms_in_timed_region": state["epoch_switch_ms"], "minibatches_per_epoch": state["nmb"],
            "gpu_prewarm": prewarm, "roofline": roofline, "step_bytes": step_bytes, "roofline_cold": cold, "roofline_mfma": mfma, "kernels": kernels, "kernels_standalone": kernels_sa,
            "kernels_source": ("HIP events on every family inside the timed region" if timers == "all" else
                               f"SCORE: HIP events inside the timed region; the other families: a second pass of {args.steps} steps of the same schedule "
                               "with every family bracketed (events cost the step 1.5-2 %, so the timed region brackets the roofline kernel only)"),
        }
        if world > 1:
            out["config"]["exchange"] = ("owner-applied update, parameter slices all-gathered in place" if getattr(loop, "owner_applied", False) else
                                         "gradient all-gather, whole-table update on every replica") if not args.partition_table else "partitioned table: owner-computes over gradient lists"
        if single_same_shape is not None:
            out["single_gpu_same_shape"] = single_same_shape
            if "value" in single_same_shape:
                out["weak_scaling_efficiency"] = (rows_total / elapsed) / (world * single_same_shape["value"])
                out["weak_scaling_efficiency_note"] = ("value / (n_gpus x single_gpu_same_shape.value), both measured in this job on the same per-GPU "
                                                       "shape; the N = 1 default line of this script is configs[2] (shorter sequences) and is NOT its denominator")
        if sweep is not None:
            sweep.append({"batch_sequences": args.batch_sequences, "interactions_per_s": rows_total / elapsed,
                          "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "interactions_per_step": rows_per_launch, "note": "the timed run"})
            out["batch_sweep"] = sorted((x for x in sweep if "batch_sequences" in x), key=lambda x: x["batch_sequences"]) + [x for x in sweep if "error" in x]
            # throughput at the largest batch with evidence of unchanged LSTM quality (tools/planted_batch_sweep.py --json): `value`
            # itself when the run is at that batch (the default), and the hardware's max-batch figure beside it
            qn = quality_neutral_batch()
            hit = qn and [x for x in out["batch_sweep"] if x.get("batch_sequences") == qn["batch_sequences"]]
            if hit and model_kind != 2:
                out["value_quality_neutral"] = {"value": hit[0]["interactions_per_s"], "unit": "interactions/s", "ms_per_step": hit[0]["ms_per_step"],
                                                "batch_sequences_per_gpu": qn["batch_sequences"], "criterion": qn["criterion"], "table": qn["table"]}
            top = [x for x in out["batch_sweep"] if x.get("batch_sequences") == MAX_BATCH]
            if top:
                out["value_max_batch"] = {"value": top[0]["interactions_per_s"], "unit": "interactions/s", "ms_per_step": top[0]["ms_per_step"],
                                          "batch_sequences_per_gpu": MAX_BATCH, "score_kernel_roofline": top[0].get("score_kernel"),
                                          "note": "two optimiser steps per epoch of this workload: the hardware's throughput regime; at equal epochs the "
                                                  "LSTM's test MRR is a quarter lower there (NOTES.md section 3), so it is not the headline"}
                if top[0].get("score_kernel") and isinstance(out.get("roofline"), dict) and args.batch_sequences != MAX_BATCH:
                    # the same kernel where its launch is long enough to be bytes-bound: at the headline's ~256 K rows a launch is
                    # ~20 us of fixed cost (dispatch, ramp, drain) + rows at 0.62 of the peak (profiles/r04_tail_experiments.md)
                    out["roofline"]["at_max_batch"] = {k: top[0]["score_kernel"][k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                                                                "algorithmic_bytes_per_launch") if k in top[0]["score_kernel"]}
                    out["roofline"]["at_max_batch"]["batch_sequences_per_gpu"] = MAX_BATCH
        if small is not None:
            out["small_steps"] = small
        if crc_ranks:
            out["param_crc"] = crc_ranks[0]
            out["param_crc_ranks_equal"] = all(c == crc_ranks[0] for c in crc_ranks)
            out["param_crc_ranks"] = len(crc_ranks)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, model_kind, loss_kind)
            except Exception as e:  # the throughput line must survive a host-side problem (e.g. memory limits)
                out["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and not args.no_mrr and not args.partition_table:
            try:  # evaluation.rs:27-41 at catalogue scale: [users x d] . [d x items] on f32 MFMA with the rank count in the epilogue
                out["mrr_gemm"] = mrr_gemm(model, args)
            except Exception as e:
                out["mrr_gemm"] = {"error": repr(e)}
        if world == 1 and not args.no_mrr:
            try:
                out["test_mrr"] = movielens_mrr()
            except Exception as e:  # the throughput line must survive a fixture problem
                out["test_mrr"] = {"error": repr(e)}
            try:  # the reference's own Criterion bench (benches/benchmark.rs:16-71; no published number): ms per fit call
                out["reference_criterion_bench"] = reference_criterion_bench()
            except Exception as e:
                out["reference_criterion_bench"] = {"error": repr(e)}
        line = json.dumps(out)
    backend.close()
    # RCCL writes a version banner through C stdio, which is only flushed at exit when stdout is a pipe:
    # every rank flushes it now, and rank 0 prints after a barrier, so the JSON line is the last line
    import ctypes

    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(line, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
