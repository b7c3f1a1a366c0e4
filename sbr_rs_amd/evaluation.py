"""Evaluation — mirror of ``sbr::evaluation`` (/root/reference/src/evaluation.rs)."""
from __future__ import annotations

from .data import CompressedInteractions


def mrr_score(model, test: CompressedInteractions) -> float:
    """MRR of the last item of each test sequence, all but the last item being the inputs
    (evaluation.rs:12-48).  Runs on the device through ``sbr_mrr_score``."""
    mrr, _ranks = model.params.mrr_score(test.user_pointers, test.item_ids)
    return mrr


def mrr_ranks(model, test: CompressedInteractions):
    """As :func:`mrr_score` but also returns the integer ranks (one per user with >= 2 items)."""
    return model.params.mrr_score(test.user_pointers, test.item_ids)
