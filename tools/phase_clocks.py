#!/usr/bin/env python
"""Where a one-sequence optimiser step spends its time inside the one-launch step runs (epoch_steps_kernel): per-phase s_memtime
ticks of the run's workgroup (sbr_fit_debug_phase_clocks), for the reference's Criterion shapes (benches/benchmark.rs:26-64) and
MovieLens-100K (BASELINE configs[1]), with the wall time per step beside them.

    python tools/phase_clocks.py [criterion|movielens]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

NAMES = ["ids+order+gather", "scan+scores", "backward scan", "reduce+update", "closing barrier"]
TICKS_PER_US = float(os.environ.get("SBR_TICKS_PER_US", "2400"))  # clock64() = s_memtime: the shader clock (measured: ~2.37 GHz on MI355X)


def run(label, model, ptr, items, epochs):
    plan = model.fit_begin(ptr, items)
    plan.steps(0, plan.epoch_prepare())  # warm-up epoch
    model.synchronize()
    c0 = plan.phase_clocks()
    t0 = time.perf_counter()
    for _ in range(epochs):
        plan.steps(0, plan.epoch_prepare())
    model.synchronize()
    dt = time.perf_counter() - t0
    c1 = plan.phase_clocks()
    steps = c1[5] - c0[5]
    per = [(a - b) / max(steps, 1) / TICKS_PER_US for a, b in zip(c1[:5], c0[:5])]
    print(f"{label}: {steps} steps, wall {1e6 * dt / max(steps, 1):.1f} us/step (incl. epoch preparation); phases us/step: "
          + ", ".join(f"{n} {v:.2f}" for n, v in zip(NAMES, per)) + f"; sum {sum(per):.2f}", flush=True)
    plan.close()


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "criterion"
    if what == "criterion":
        import criterion_bench as cb

        data = cb.sample_data(10_000)
        for kind in ("lstm", "ewma"):
            m = cb.build(kind, data.num_items(), 1)
            run(f"criterion {kind}", m.params, data.user_pointers, data.item_ids, 3)
    else:
        from helpers import movielens_protocol
        from sbr_rs_amd._abi import make_hparams
        from sbr_rs_amd.engine import Model

        data, train, _test, rng = movielens_protocol()
        hp = make_hparams(data.num_items(), 128, 32, 0.16, 0.0004, 2, 1, 0, 1, rng.state_seed(), 1, 1, 0, 1)
        run("movielens ewma hinge", Model(hp), train.user_pointers, train.item_ids, 3)


if __name__ == "__main__":
    main()
