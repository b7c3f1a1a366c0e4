"""Error types of the boundary — mirror of /root/reference/src/lib.rs:84-97."""
from __future__ import annotations


class EngineError(RuntimeError):
    """ABI-level failure (invalid argument, HIP error, no device ...)."""

    def __init__(self, status, message):
        self.status = status
        super().__init__(f"{status.name}: {message}")


class FittingError(Exception):
    """Fitting error types (lib.rs:92-97).  ``FittingError.NoInteractions`` is the variant."""


class _NoInteractions(FittingError):
    """No interactions were given."""


FittingError.NoInteractions = _NoInteractions


class PredictionError(Exception):
    """Prediction error types (lib.rs:84-89).  ``PredictionError.InvalidPredictionValue`` is the variant."""


class _InvalidPredictionValue(PredictionError):
    """Failed prediction due to numerical issues."""


PredictionError.InvalidPredictionValue = _InvalidPredictionValue
