"""tests/golden/reference_order_*.npz — the engine-side dumps of the five MovieLens protocol cases in REFERENCE ORDER that a
maintainer with cargo compares the crate against (integration/rust_check/, tools/compare_with_crate.py) — are reproduced by
the oracle's reference-order mode (CPU) and by the ENGINE's (GPU: sbr_model_set_reference_order, sbr_group_fit with the
replicas in reference order), field by field, bit for bit.  ≙ /root/reference/src/models/lstm.rs:427-520, ewma.rs:463-507,
sequence_model.rs:76-98, :109, :137."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

import make_reference_order_golden as gold  # noqa: E402


def _check(name, got):
    want = np.load(gold.case_file(name))
    assert sorted(want.files) == sorted(got)
    for k in want.files:
        a, b = np.asarray(want[k]), np.asarray(got[k])
        assert a.shape == b.shape and a.dtype == b.dtype, (name, k)
        if k == "fit_loss_true":  # the mean loss is accumulated order-free in f64 on the device: tolerance
            assert float(b) == pytest.approx(float(a), rel=1e-6), (name, k)
        else:
            assert a.tobytes() == b.tobytes(), (name, k)


def test_streams_and_split_are_the_committed_ones(oracle_lib):
    want, got = np.load(os.path.join(gold.GOLDEN, "reference_order_streams.npz")), gold.streams()
    assert sorted(want.files) == sorted(got)
    for k in want.files:
        assert np.asarray(want[k]).tobytes() == np.asarray(got[k]).tobytes(), k
    # the split the protocol rests on (data.rs:69-88 on the fixture): one user in five held out, nobody in both
    assert int(want["train_interactions"]) + int(want["test_interactions"]) == 100_000
    assert 0.15 < int(want["test_users_with_data"]) / (int(want["test_users_with_data"]) + int(want["train_users_with_data"])) < 0.25


@pytest.mark.parametrize("name,kind,loss,threads", gold.CASES)
def test_oracle_reproduces_the_committed_dump(oracle_lib, name, kind, loss, threads):
    from oracle.oracle import OracleModel

    def make(hp):
        m = OracleModel(hp)
        m.set_reference_order(True)
        return m

    _check(name, gold.build_case(name, kind, loss, threads, make))


class _Group:
    """The single-process group (one replica per worker) behind the surface build_case drives."""

    class _Plan:
        def __init__(self, gp):
            self.gp = gp

        def epoch_prepare(self):
            return self.gp.epoch_prepare()

        def step(self, mb):
            self.gp.step(mb)

        def minibatch_rows(self, mb, q=0):
            return self.gp.member(q).minibatch_rows(mb)

        def debug_fetch(self, which, rows, q=0):
            return self.gp.member(q).debug_fetch(which, rows)

        def close(self):
            self.gp.close()

    def __init__(self, hp, world):
        from sbr_rs_amd.engine import group_create

        self.models = group_create(hp, world)
        for m in self.models:
            m.set_reference_order(True)

    def get_rng(self):
        return self.models[0].get_rng()

    def fit_begin(self, ptr, items):
        from sbr_rs_amd.engine import GroupPlan

        return self._Plan(GroupPlan(self.models, ptr, items))

    def fit(self, ptr, items):
        from sbr_rs_amd.engine import group_fit

        return group_fit(self.models, ptr, items)

    def mrr_score(self, ptr, items):
        return self.models[-1].mrr_score(ptr, items)

    def last_fit_lagged_loss(self):
        return self.models[0].last_fit_lagged_loss()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,loss,threads", gold.CASES)
def test_engine_reproduces_the_committed_dump(name, kind, loss, threads):
    from sbr_rs_amd.engine import Model

    def make(hp):
        if threads > 1:
            return _Group(hp, threads)
        m = Model(hp)
        m.set_reference_order(True)
        return m

    _check(name, gold.build_case(name, kind, loss, threads, make))


def test_comparison_tool_finds_the_first_divergence(tmp_path):
    """tools/compare_with_crate.py on a dump played back from the golden files (exit 0, no DIFFERS), and on one with a single
    visiting-order entry changed (exit 1, the line names that stream and that element)."""
    import json
    import subprocess

    g = np.load(os.path.join(gold.GOLDEN, "reference_order_streams.npz"))
    dump = {"crate_name": "golden files played back",
            "streams": {k: [int(x) for x in g[k]] for k in ("next_u32", "uniform_u64", "uniform_usize_1683", "shuffle_10", "normal_bits", "gen_seed16")},
            "split": {k: int(g[k]) for k in ("train_users_with_data", "test_users_with_data", "train_interactions", "test_interactions",
                                              "train_items_fnv", "test_items_fnv")},
            "cases": []}
    for name, _kind, _loss, _threads in gold.CASES:
        o = np.load(gold.case_file(name))
        dump["cases"].append({"name": name, "fit_loss": float(o["fit_loss_lagged"]), "test_mrr": float(o["test_mrr"]), "train_mrr": 0.0,
                              "test_ranks": [int(x) for x in o["test_ranks"]],
                              "replay": {"assumed_wyrm_lstm_draws": 8192 if name.startswith("lstm") else 0, "num_subsequences": int(o["num_subsequences"]),
                                         "shuffled_order": o["shuffled_order"].tolist(), "worker_seeds": o["worker_seeds"].tolist(),
                                         "first_epoch_order": o["first_epoch_order"].tolist(), "first_epoch_raw_draws": o["first_epoch_raw_draws"].tolist()}})
    tool = os.path.join(os.path.dirname(gold.__file__), "compare_with_crate.py")
    good = tmp_path / "good.json"
    good.write_text(json.dumps(dump))
    res = subprocess.run([sys.executable, tool, str(good)], capture_output=True, text=True)
    assert res.returncode == 0 and "DIFFERS" not in res.stdout, res.stdout[-2000:]
    dump["cases"][3]["replay"]["first_epoch_order"][0][17][0] += 1
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(dump))
    res = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert res.returncode == 1 and "DIFFERS  first epoch's visiting order (:109): first at element 34" in res.stdout, res.stdout[-2000:]
