"""EWMA sequence model — host-side mirror of ``sbr::models::ewma``
(/root/reference/src/models/ewma.rs).  The state recurrence the engine runs is the code's
(ewma.rs:302-313), not the module doc formula (ewma.rs:11), which is wrong in the reference.
The reference's unused ``fc1``/``fc2`` parameters (ewma.rs:179-188) are not reproduced."""
from __future__ import annotations

from ._abi import ModelKind
from .engine import Model
from .lstm import _HyperparametersBase, _ImplicitSequenceModel
from .models import Parallelism
from .rng import XorShiftRng


class Hyperparameters(_HyperparametersBase):
    """Hyperparameters describing the EWMA model (ewma.rs:45-206)."""

    @classmethod
    def new(cls, num_items: int, max_sequence_length: int) -> "Hyperparameters":
        return cls(num_items, max_sequence_length)

    @classmethod
    def random(cls, num_items: int, rng: XorShiftRng) -> "Hyperparameters":
        h, uni = cls._random_common(num_items, rng)
        h._parallelism = Parallelism.Asynchronous if uni(0.0, 1.0) < 0.5 else Parallelism.Synchronous
        h._num_epochs = 2 ** (3 + rng.below(4))
        return h

    def build(self, device_rank: int = 0) -> "ImplicitEWMAModel":
        """Build the implicit EWMA model (ewma.rs:201-205)."""
        return ImplicitEWMAModel(*self._build_engine(int(ModelKind.EWMA), device_rank))


class ImplicitEWMAModel(_ImplicitSequenceModel):
    """Implicit EWMA model (ewma.rs:401-429)."""
