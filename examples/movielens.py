"""The crate's README example (src/lib.rs:22-58) against the Python host layer of the MI355X engine.

    python -m sbr_rs_amd.build
    python examples/movielens.py [data.csv]       # default: the fixture under tests/golden
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sbr_rs_amd as sbr  # noqa: E402

data = sbr.datasets.download_movielens_100k(sys.argv[1] if len(sys.argv) > 1 else None)
rng = sbr.XorShiftRng.from_seed(bytes([42] * 16))
train, test = sbr.data.user_based_split(data, rng, 0.2)
train_mat, test_mat = train.to_compressed(), test.to_compressed()
print(f"Train: {train.len()}, test: {test.len()}")

model = (sbr.lstm.Hyperparameters.new(data.num_items(), 32)
         .embedding_dim(32).learning_rate(0.16).l2_penalty(0.0004)
         .lstm_variant(sbr.LSTMVariant.Normal).loss(sbr.Loss.WARP).optimizer(sbr.Optimizer.Adagrad)
         .num_epochs(10).rng(rng).build())
start = time.perf_counter()
loss = model.fit(train_mat)
elapsed = time.perf_counter() - start
print(f"Train MRR {sbr.evaluation.mrr_score(model, train_mat):.4f} at loss {loss:.4f} and "
      f"test MRR {sbr.evaluation.mrr_score(model, test_mat):.4f} (in {elapsed:.3f} s)")
user = model.user_representation([50, 181, 258])
print("scores:", model.predict(user, [1, 100, 300]))
sbr.persistence.save_model(model, "/tmp/sbr_movielens_model.npz")  # parameters + optimiser state + counters
