"""sbr-rs_amd — MI355X-native engine for sbr's sequence-recommender hot path.

Python host side mirroring the reference crate's public modules (``sbr::data``,
``sbr::models::{lstm, ewma}``, ``sbr::evaluation``); all compute goes through the C-ABI of
libsbr_hip.so (include/sbr_hip.h) into hand-written gfx950 kernels.
"""
from . import data, datasets, evaluation, ewma, lstm, models, persistence  # noqa: F401
from .errors import EngineError, FittingError, PredictionError  # noqa: F401
from .models import ImplicitUser, Loss, LSTMVariant, Optimizer, Parallelism  # noqa: F401
from .rng import XorShiftRng  # noqa: F401

__all__ = ["data", "datasets", "persistence", "evaluation", "ewma", "lstm", "models", "Loss", "Optimizer", "Parallelism", "LSTMVariant",
           "ImplicitUser", "FittingError", "PredictionError", "EngineError", "XorShiftRng"]
