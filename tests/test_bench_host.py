"""CPU: the host-side bookkeeping of bench.py — which BASELINE.json config a run's shape is labelled as, when the measured
HBM traffic of the score kernel may be printed, and the committed quality-neutral batch."""
import json
import os
import types

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    base = dict(model="lstm", loss="warp", dim=128, items=1_000_000, users=100_000, max_len=64, partition_table=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_workload_labels_follow_baseline_configs():
    assert bench.workload_label(_args(), 1) == "BASELINE.json configs[2]"
    assert bench.workload_label(_args(users=125_000, max_len=128), 8) == "BASELINE.json configs[3]"
    assert "configs[3]'s per-GPU shape at 4 GPUs" in bench.workload_label(_args(users=125_000, max_len=128), 4)
    # configs[2]'s shape on several GPUs is not a BASELINE config; configs[3]'s per-GPU shape on one GPU is the denominator of the
    # weak-scaling efficiency (--scale-shape) and says so
    assert bench.workload_label(_args(), 8) == "custom workload"
    assert "same-shape denominator" in bench.workload_label(_args(users=125_000, max_len=128), 1)
    c4 = _args(model="ewma", loss="hinge", dim=256, items=10_000_000, users=125_000, max_len=128, partition_table=True)
    assert bench.workload_label(c4, 8) == "BASELINE.json configs[4]"
    assert "configs[4]'s shape" in bench.workload_label(c4, 2)
    assert bench.workload_label(_args(dim=64), 1) == "custom workload"


def test_profiled_traffic_is_printed_only_at_a_profiled_operating_point():
    """bench.py --traffic profile (the fallback of --traffic live): profiles/score_kernel_traffic.json holds one entry per profiled
    operating point; an entry is used only for a run at that point — same dim and table, rows per launch and negatives per row
    within 5 %."""
    prof = json.load(open(os.path.join(ROOT, "profiles", "score_kernel_traffic.json")))
    for which in ("warm", "cold"):
        assert len(prof[which]) >= 2
        for e in prof[which]:
            d, items = e.get("dim", 128), e.get("items")
            up, lo = bench.measured_traffic(which, e["rows_per_launch"], e["mean_negatives_scored"], d, items)
            assert up is not None and lo < up
            assert 1.0 < lo / e["algorithmic_bytes_per_launch"] < up / e["algorithmic_bytes_per_launch"] < 1.4
            assert bench.measured_traffic(which, e["rows_per_launch"] * 1.08, e["mean_negatives_scored"] * 1.2, d, items) == (None, None)
            assert bench.measured_traffic(which, e["rows_per_launch"], e["mean_negatives_scored"], 64, items) == (None, None)
            assert bench.measured_traffic(which, e["rows_per_launch"], e["mean_negatives_scored"], d, 12345) == (None, None)
    # the headline's operating point (8 192 sequences per step) and the max-batch one (50 000) are both on file
    rows = sorted(e["rows_per_launch"] for e in prof["warm"] if e.get("dim", 128) == 128)
    assert rows[0] < 300_000 and rows[-1] > 1_500_000
    assert bench.measured_traffic("absent", 1.0, 1.0, 128) == (None, None)


def test_quality_neutral_batch_is_committed_with_its_evidence():
    qn = bench.quality_neutral_batch()
    assert qn and qn["batch_sequences"] == 8192
    rows = {r["batch_sequences"]: r["mrr_mean"] for r in qn["rows"]}
    base = rows[qn["reference_batch"]]
    assert all(rows[b] >= 0.97 * base for b in rows if b <= qn["batch_sequences"])
    assert rows[16384] < 0.97 * base and rows[50000] < 0.8 * base  # what the bench's default batch costs the LSTM
    assert os.path.exists(os.path.join(ROOT, qn["table"]))


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment — the driver's command shape — must start two ranks
    through torch.distributed.run by itself.  On this CPU box each rank then stops at the engine's "needs an MI355X" check:
    what is asserted here is that both ranks were started (round 3 exited before doing anything)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=600)
    import torch

    if torch.cuda.is_available():
        assert res.returncode == 0, res.stderr[-2000:]
        assert json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert "must be launched through" not in res.stderr
        assert res.stderr.count("bench.py needs an MI355X") == 2, res.stderr[-2000:]


def _trace_db(path, dispatches):
    """A rocpd-shaped database: `kernels` (name, start, duration) and `counters_collection` (dispatch_id, kernel_name, counter_name,
    value) with the given (kernel name, duration ns, FETCH_SIZE) dispatches in order."""
    import sqlite3

    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer, duration integer)")
    db.execute("create table counters_collection (dispatch_id integer, kernel_name text, counter_name text, value real)")
    t = 0
    for i, (name, dur, fetch) in enumerate(dispatches):
        db.execute("insert into kernels values (?, ?, ?, ?)", (name, t, t + dur, dur))
        for inst in range(2):  # two counter instances per dispatch: the tools sum them
            db.execute("insert into counters_collection values (?, ?, 'FETCH_SIZE', ?)", (i, name, fetch / 2))
        t += dur + 10
    db.commit()
    db.close()


def test_pmc_step_window_and_timed_region_table(tmp_path, capsys):
    """tools/pmc_dispatches.step_window (bench.py's `roofline.traffic_step`) and tools/rocpd_stats.py --window (the timed region's own
    kernel table): only the dispatches from the first-th up to the (first + count)-th dispatch of the marking kernel are counted."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_dispatches
    import rocpd_stats

    step = [("void sbr::ewma_seq_kernel<256, false>(sbr::ModelView)", 2000, 100.0), ("void sbr::seg_short_kernel<256, sbr::EmitApply, false>(x)", 3000, 200.0),
            ("radix_hist_kernel", 100, 4.0)]
    disp = [("warm_up_only_kernel", 50, 1000.0)] + step * 5  # five steps after one unrelated dispatch
    path = str(tmp_path / "run_results.db")
    _trace_db(path, disp)
    assert pmc_dispatches.per_dispatch(path, "FETCH_SIZE", "ewma_seq_kernel") == [100.0] * 5
    win = pmc_dispatches.step_window(path, "FETCH_SIZE", "ewma_seq_kernel", 1, 3)  # steps 1, 2, 3
    assert win == {"ewma_seq_kernel": 300.0, "seg_short_kernel": 600.0, "radix_hist_kernel": 12.0}
    assert pmc_dispatches.step_window(path, "FETCH_SIZE", "ewma_seq_kernel", 2, 3)["seg_short_kernel"] == 600.0  # to the end of the trace
    assert pmc_dispatches.step_window(path, "FETCH_SIZE", "ewma_seq_kernel", 4, 3) is None  # not that many steps
    sys.argv = ["rocpd_stats.py", path, str(tmp_path / "out.md"), "--window", "ewma_seq_kernel", "1", "3"]
    rocpd_stats.main()
    text = open(tmp_path / "out.md").read()
    whole, timed = text.split("Timed region only")
    assert "| `ewma_seq_kernel<256, false>` | 5 | 0.010 | 2.00 |" in whole and "warm_up_only_kernel" in whole
    assert "| `ewma_seq_kernel<256, false>` | 3 | 0.006 | 2.00 |" in timed and "warm_up_only_kernel" not in timed
    capsys.readouterr()
