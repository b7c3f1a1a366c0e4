"""Model enums — mirror of /root/reference/src/models/mod.rs:9-41 and lstm.rs:28-35."""
from __future__ import annotations

import enum
from dataclasses import dataclass

import numpy as np


class Loss(enum.IntEnum):
    """The loss used for training the model (mod.rs:15-23)."""

    BPR = 0
    Hinge = 1
    WARP = 2


class Optimizer(enum.IntEnum):
    """Optimizer used to train the model (mod.rs:26-32)."""

    Adagrad = 0
    Adam = 1


class Parallelism(enum.IntEnum):
    """Type of parallelism (mod.rs:35-41).  Synchronous = every device sees every update before its
    next minibatch (deterministic owner-reduce rendezvous).  Asynchronous = the deterministic analogue
    of Hogwild: with more than one device, minibatch k+1 is computed on parameters that lack update k
    (staleness exactly one step), which hides the exchange under the computation."""

    Asynchronous = 0
    Synchronous = 1


class LSTMVariant(enum.IntEnum):
    """Type of LSTM layer (lstm.rs:28-35)."""

    Normal = 0
    Coupled = 1


@dataclass
class ImplicitUser:
    """The user representation used by implicit sequence models (mod.rs:9-12)."""

    user_embedding: np.ndarray
