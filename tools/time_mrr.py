"""Times sbr_mrr_score (MFMA scoring + rank kernel) at catalogue scale: U users x 1M items, dim 128."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import hparams, synthetic_interactions
from sbr_rs_amd.engine import Model

U, I, D = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 1_000_000, 128
for kind, name in ((2, "EWMA"), (0, "LSTM")):
    m = Model(hparams(I, 64, D, kind, 2, B=1024))
    ptr, it = synthetic_interactions(U, I, 40, seed=5, min_len=2)
    m.mrr_score(ptr[:129], it[: int(ptr[128])])  # warm-up
    m.timing_enable(True); m.timing_read()
    t0 = time.perf_counter()
    mrr, ranks = m.mrr_score(ptr, it)
    dt = time.perf_counter() - t0
    t = m.timing_read()
    rank_ms, n = t["RANK"]
    flops = 2.0 * len(ranks) * I * D
    print(f"{name}: {len(ranks)} users x {I} items x dim {D}: mrr {mrr:.6f} wall {dt*1e3:.1f} ms; rank kernels {rank_ms:.2f} ms "
          f"({n} chunks) = {flops / (rank_ms * 1e-3) / 1e12:.1f} TFLOP/s f32 MFMA; forward {t['RECURRENT_FWD'][0]:.2f} ms")
