"""Independent check of the oracle's forward/backward against torch float64 autograd.

The oracle restates wyrm's reverse-mode pass by hand (loss.backward(1.0),
/root/reference/src/models/sequence_model.rs:160-161); this test rebuilds the same graph
(lstm.rs:258-337 / ewma.rs:266-352) with torch ops in float64 and lets autograd differentiate it.
"""
import numpy as np
import pytest
import torch

from helpers import LOSS_BPR, LOSS_HINGE, LOSS_WARP, hparams, synthetic_interactions
from oracle.oracle import OracleModel
from sbr_rs_amd._abi import Debug, ModelKind, Param


def _torch_reference(model_kind, loss_kind, d, params, seqs, negs):
    """seqs: list of item-id lists; negs: list of per-step negative ids.  Returns loss, grads."""
    E = torch.tensor(params["E"], dtype=torch.float64).requires_grad_(True)
    b = torch.tensor(params["b"], dtype=torch.float64).requires_grad_(True)
    leaves = [E, b]
    if model_kind == ModelKind.EWMA:
        alpha = torch.tensor(params["alpha"], dtype=torch.float64).requires_grad_(True)
        leaves.append(alpha)
    else:
        W = torch.tensor(params["W"], dtype=torch.float64).requires_grad_(True)
        bW = torch.tensor(params["bW"], dtype=torch.float64).requires_grad_(True)
        leaves += [W, bW]
    total = torch.zeros((), dtype=torch.float64)
    hiddens, xs = [], []
    for items, neg in zip(seqs, negs):
        h = torch.zeros(d, dtype=torch.float64)
        c = torch.zeros(d, dtype=torch.float64)
        for t in range(len(items) - 1):
            x = E[items[t]]
            x.retain_grad()
            xs.append(x)
            if model_kind == ModelKind.EWMA:
                a = torch.sigmoid(alpha)
                h = x if t == 0 else a * h + (1 - a) * x
            else:
                z = torch.cat([x, h]) @ W + bW
                if model_kind == ModelKind.LSTM_NORMAL:
                    i, f, g, o = torch.sigmoid(z[:d]), torch.sigmoid(z[d:2 * d]), torch.tanh(z[2 * d:3 * d]), torch.sigmoid(z[3 * d:])
                else:
                    f, g, o = torch.sigmoid(z[:d]), torch.tanh(z[d:2 * d]), torch.sigmoid(z[2 * d:])
                    i = 1 - f
                c = f * c + i * g
                h = o * torch.tanh(c)
            hiddens.append(h)
            pos = h @ E[items[t + 1]] + b[items[t + 1]]
            ng = h @ E[neg[t]] + b[neg[t]]
            if loss_kind == LOSS_BPR:
                total = total + torch.sigmoid(ng - pos)
            else:
                total = total + torch.relu(1 + ng - pos)
    total.backward()
    return total.item(), leaves, xs, hiddens


@pytest.mark.parametrize("model_kind", [ModelKind.LSTM_NORMAL, ModelKind.LSTM_COUPLED, ModelKind.EWMA])
@pytest.mark.parametrize("loss_kind", [LOSS_HINGE, LOSS_BPR, LOSS_WARP])
def test_oracle_gradients_match_autograd(oracle_lib, model_kind, loss_kind):
    I, d, T, B = 60, 16, 9, 5
    ptr, items = synthetic_interactions(12, I, T, seed=3)
    hp = hparams(I, T, d, int(model_kind), loss_kind, epochs=1, B=B, l2=0.0)
    m = OracleModel(hp)
    rs = np.random.RandomState(0)
    # move off the init so gates/biases are non-trivial
    E = (rs.randn(I, d) * 0.5).astype(np.float32)
    b = (rs.randn(I) * 0.3).astype(np.float32)
    m.set_param(Param.ITEM_EMBEDDING, E)
    m.set_param(Param.ITEM_BIAS, b)
    params = {"E": E, "b": b}
    if model_kind == ModelKind.EWMA:
        alpha = (rs.randn(d) * 0.7).astype(np.float32)
        m.set_param(Param.EWMA_ALPHA, alpha)
        params["alpha"] = alpha
    else:
        ng = 4 if model_kind == ModelKind.LSTM_NORMAL else 3
        W = (rs.randn(2 * d, ng * d) * 0.3).astype(np.float32)
        bW = (rs.randn(ng * d) * 0.2).astype(np.float32)
        m.set_param(Param.LSTM_W, W)
        m.set_param(Param.LSTM_B, bW)
        params["W"], params["bW"] = W, bW
    plan = m.fit_begin(ptr, items)
    nmb = plan.epoch_prepare()
    assert nmb >= 1
    R = plan.minibatch_rows(0)
    plan.step_local(0)
    in_idx = plan.debug_fetch(Debug.IN_IDX, R)
    out_idx = plan.debug_fetch(Debug.OUT_IDX, R)
    neg = plan.debug_fetch(Debug.NEGATIVES, R)
    H = plan.debug_fetch(Debug.HIDDEN, R)
    loss = plan.debug_fetch(Debug.LOSS, R)
    dX = plan.debug_fetch(Debug.DINPUT, R)
    dense = plan.debug_fetch(Debug.DENSE_GRAD, R)

    # Rebuild the sequences from the packed time-major rows: row(t, b) = off[t] + b, sequences in
    # length-descending order; step-t rows are a prefix of the step-(t-1) sequences.
    seq_rows = []
    seqs_sorted = _minibatch_sequences(ptr, items, T, hp, m, plan)
    nb = len(seqs_sorted)
    maxlen = max(len(s) for s in seqs_sorted)
    off = [0]
    for t in range(maxlen - 1):
        off.append(off[-1] + sum(1 for s in seqs_sorted if len(s) - 1 > t))
    assert off[-1] == R
    negs = []
    for bi, s in enumerate(seqs_sorted):
        rows = [off[t] + bi for t in range(len(s) - 1)]
        assert [int(in_idx[r]) for r in rows] == s[:-1]
        assert [int(out_idx[r]) for r in rows] == s[1:]
        negs.append([int(neg[r]) for r in rows])
        seq_rows.append(rows)

    total, leaves, xs, hiddens = _torch_reference(model_kind, loss_kind, d, params, seqs_sorted, negs)
    np.testing.assert_allclose(loss.sum(dtype=np.float64), total, rtol=2e-5, atol=1e-5)
    flat_rows = [r for rows in seq_rows for r in rows]
    Href = np.stack([h.detach().numpy() for h in hiddens])
    np.testing.assert_allclose(H[flat_rows], Href, rtol=1e-4, atol=2e-6)
    dXref = np.stack([x.grad.numpy() if x.grad is not None else np.zeros(d) for x in xs])
    # x.grad of the gathered row only holds the path through the recurrence (the target/negative
    # gathers are separate index nodes), which is exactly the oracle's dX.
    np.testing.assert_allclose(dX[flat_rows], dXref, rtol=2e-4, atol=2e-5)
    if model_kind == ModelKind.EWMA:
        np.testing.assert_allclose(dense, leaves[2].grad.numpy(), rtol=2e-4, atol=2e-5)
    else:
        ng = 4 if model_kind == ModelKind.LSTM_NORMAL else 3
        dW = dense[: 2 * d * ng * d].reshape(2 * d, ng * d)
        dbW = dense[2 * d * ng * d:]
        np.testing.assert_allclose(dW, leaves[2].grad.numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(dbW, leaves[3].grad.numpy(), rtol=2e-4, atol=2e-5)


def _minibatch_sequences(ptr, items, T, hp, m, plan):
    """Sequences of minibatch 0 in the oracle's packed order, recovered from its index rows."""
    R = plan.minibatch_rows(0)
    in_idx = plan.debug_fetch(Debug.IN_IDX, R)
    out_idx = plan.debug_fetch(Debug.OUT_IDX, R)
    # B_0 = number of rows at t = 0.  Step-t rows are a prefix of step-(t-1) sequences, and
    # in_idx[t+1][b] == out_idx[t][b]; walk greedily.
    B = int(hp.batch_sequences)
    # candidate B_0 values: any nb <= B such that chains are consistent and total rows == R
    for nb in range(min(B, R), 0, -1):
        seqs = [[int(in_idx[b])] for b in range(nb)]
        last_out = [int(out_idx[b]) for b in range(nb)]
        alive = nb
        pos = nb
        ok = True
        while pos < R:
            # next step has bt <= alive rows
            bt = 0
            while bt < alive and pos + bt < R and int(in_idx[pos + bt]) == last_out[bt]:
                bt += 1
            # bt might be over-counted by coincidence only if ids repeat; accept greedy
            if bt == 0:
                ok = False
                break
            for b in range(bt):
                seqs[b].append(last_out[b])
                last_out[b] = int(out_idx[pos + b])
            alive = bt
            pos += bt
        if ok and pos == R:
            for b in range(nb):
                seqs[b].append(last_out[b])
            if all(len(s) <= T for s in seqs) and sorted((len(s) for s in seqs), reverse=True) == [len(s) for s in seqs]:
                return seqs
    raise AssertionError("could not recover minibatch structure")
