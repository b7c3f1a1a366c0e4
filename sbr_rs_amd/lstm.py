"""LSTM-based sequence model — host-side mirror of ``sbr::models::lstm``
(/root/reference/src/models/lstm.rs): same builder, same method names, same error behaviour,
calling the gfx950 engine through the C-ABI instead of wyrm."""
from __future__ import annotations

import os

import numpy as np

from ._abi import ModelKind, make_hparams
from .data import CompressedInteractions
from .engine import Model
from .models import ImplicitUser, Loss, LSTMVariant, Optimizer, Parallelism
from .rng import XorShiftRng


class _HyperparametersBase:
    """Builder shared by lstm::Hyperparameters (lstm.rs:39-202) and ewma::Hyperparameters
    (ewma.rs:45-206).  Defaults follow ``new`` (lstm.rs:56-71 / ewma.rs:61-75)."""

    def __init__(self, num_items: int, max_sequence_length: int):
        self._num_items = int(num_items)
        self._max_sequence_length = int(max_sequence_length)
        self._item_embedding_dim = 16
        self._learning_rate = 0.01
        self._l2_penalty = 0.0
        self._loss = Loss.BPR
        self._optimizer = Optimizer.Adam
        self._parallelism = Parallelism.Synchronous
        self._rng = XorShiftRng.from_seed(os.urandom(16))  # rand::thread_rng().gen()
        self._num_threads = 1  # devices; the reference default is rayon::current_num_threads()
        self._num_epochs = 10
        self._batch_sequences = 32  # GPU minibatch (no reference counterpart; 1 = per-sequence SGD)

    def learning_rate(self, learning_rate: float):
        self._learning_rate = float(learning_rate)
        return self

    def l2_penalty(self, l2_penalty: float):
        self._l2_penalty = float(l2_penalty)
        return self

    def embedding_dim(self, embedding_dim: int):
        self._item_embedding_dim = int(embedding_dim)
        return self

    def num_epochs(self, num_epochs: int):
        self._num_epochs = int(num_epochs)
        return self

    def loss(self, loss: Loss):
        self._loss = Loss(loss)
        return self

    def num_threads(self, num_threads: int):
        """Number of parallel workers = devices (lstm.rs:110-113)."""
        self._num_threads = int(num_threads)
        return self

    def parallelism(self, parallelism: Parallelism):
        self._parallelism = Parallelism(parallelism)
        return self

    def rng(self, rng: XorShiftRng):
        self._rng = rng
        return self

    def from_seed(self, seed):
        self._rng = XorShiftRng.from_seed(seed)
        return self

    def optimizer(self, optimizer: Optimizer):
        self._optimizer = Optimizer(optimizer)
        return self

    def batch_sequences(self, batch_sequences: int):
        """Subsequences per optimiser step (engine extension)."""
        self._batch_sequences = int(batch_sequences)
        return self

    def partition_item_table(self, flag: bool = True):
        """With ``num_threads(n) > 1`` in one process: store the item table once, row range r on replica
        r's device, instead of n full copies (engine extension; results are identical)."""
        self._partition = bool(flag)
        return self

    def _build_engine(self, model_kind: int, device_rank: int):
        """Model handle(s) for build(): one handle, or the whole single-process group when the item
        table is partitioned (the replicas must then be created together)."""
        hp = self._hparams(model_kind, device_rank)
        if getattr(self, "_partition", False):
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() == self._num_threads > 1:
                # one process per GPU: this process owns the rows of its rank, the peers' rows are mapped
                from .partitioned import create_partitioned_model

                return create_partitioned_model(self._hparams(model_kind, dist.get_rank())), None
            from .engine import group_create

            group = group_create(hp, self._num_threads, partition_item_table=True)
            return group[0], group
        return Model(hp), None

    def _hparams(self, model_kind: int, device_rank: int = 0):
        return make_hparams(self._num_items, self._max_sequence_length, self._item_embedding_dim, self._learning_rate,
                            self._l2_penalty, model_kind, int(self._loss), int(self._optimizer), int(self._parallelism),
                            self._rng.state_seed(), self._num_epochs, self._num_threads, device_rank,
                            self._batch_sequences)

    @classmethod
    def _random_common(cls, num_items: int, rng: XorShiftRng):
        """``Hyperparameters::random`` (lstm.rs:141-172): same ranges; the draws come from this
        engine's RNG, so the sampled points differ from the Rust crate's."""
        def uni(lo, hi):
            return lo + (hi - lo) * rng.unit()

        h = cls(num_items, 2 ** (4 + rng.below(4)))
        h._item_embedding_dim = 2 ** (4 + rng.below(4))
        h._learning_rate = float(np.float32(10.0) ** np.float32(uni(-3.0, 0.5)))
        h._l2_penalty = float(np.float32(10.0) ** np.float32(uni(-7.0, -3.0)))
        h._loss = Loss.BPR if uni(0.0, 1.0) < 0.5 else Loss.Hinge
        h._optimizer = Optimizer.Adam if uni(0.0, 1.0) < 0.5 else Optimizer.Adagrad
        return h, uni


class Hyperparameters(_HyperparametersBase):
    """Hyperparameters for the ImplicitLSTMModel (lstm.rs:39-202)."""

    def __init__(self, num_items: int, max_sequence_length: int):
        super().__init__(num_items, max_sequence_length)
        self._lstm_type = LSTMVariant.Coupled

    @classmethod
    def new(cls, num_items: int, max_sequence_length: int) -> "Hyperparameters":
        return cls(num_items, max_sequence_length)

    def lstm_variant(self, variant: LSTMVariant):
        self._lstm_type = LSTMVariant(variant)
        return self

    @classmethod
    def random(cls, num_items: int, rng: XorShiftRng) -> "Hyperparameters":
        h, uni = cls._random_common(num_items, rng)
        h._lstm_type = LSTMVariant.Normal if uni(0.0, 1.0) < 0.5 else LSTMVariant.Coupled
        h._parallelism = Parallelism.Asynchronous if uni(0.0, 1.0) < 0.5 else Parallelism.Synchronous
        h._num_epochs = 2 ** (3 + rng.below(4))
        return h

    def build(self, device_rank: int = 0) -> "ImplicitLSTMModel":
        """Build a model out of the chosen hyperparameters (lstm.rs:197-201): parameters are
        initialised from the builder's RNG on the device."""
        kind = ModelKind.LSTM_NORMAL if self._lstm_type == LSTMVariant.Normal else ModelKind.LSTM_COUPLED
        return ImplicitLSTMModel(*self._build_engine(int(kind), device_rank))


class _ImplicitSequenceModel:
    """fit / OnlineRankingModel surface shared by both models (lstm.rs:391-416, ewma.rs:404-429)."""

    def __init__(self, engine_model: Model, group=None):
        self.params = engine_model
        self._peers = group[1:] if group else None

    def fit(self, interactions: CompressedInteractions) -> float:
        """Fit the model; returns the loss value.  Raises FittingError.NoInteractions
        (lstm.rs:395-397 → sequence_model.rs:86-88)."""
        world = int(self.params.hp.num_devices)
        if world == 1:
            return self.params.fit(interactions.user_pointers, interactions.item_ids)
        try:  # torch is needed only for the one-process-per-GPU driver
            import torch.distributed as dist

            launched = dist.is_available() and dist.is_initialized()
        except ImportError:
            launched = False
        if launched and getattr(self, "_peers", None) is None:
            # one process per GPU (torchrun): this process drives replica hp.device_rank
            if self.params.is_partitioned():
                from .partitioned import fit_partitioned

                return fit_partitioned(self.params, interactions)
            from .distributed import fit_distributed

            return fit_distributed(self.params, interactions)
        # one process, num_threads replicas (≙ the reference's rayon workers): sbr_group_fit
        from .engine import group_fit

        return group_fit(self._replicas(), interactions.user_pointers, interactions.item_ids)

    def set_reference_order(self, on: bool = True):
        """The crate's own ORDER of work at one subsequence per step (``batch_sequences(1)``, embedding_dim <= 32): negatives from the
        worker's sequential XorShiftRng stream (sequence_model.rs:58-65, :137) and, with ``num_threads(n)``, one optimiser application
        per worker in worker order (:163-166).  Call before ``fit``."""
        for m in self._replicas():
            m.set_reference_order(on)
        return self

    def last_fit_lagged_loss(self) -> float:
        """The number the reference's ``fit`` would have returned for the last single-process ``fit`` (one device or
        ``num_threads`` replicas): sequence_model.rs:157 reads the loss node before :160 runs its forward pass, so every
        subsequence of s steps contributes the running loss sum L_{s-1} of the worker's most recent earlier subsequence with at
        least s steps (the loss nodes are shared running sums, lstm.rs:322-328).  ``fit`` returns the
        true mean loss."""
        return self.params.last_fit_lagged_loss()

    def _replicas(self):
        """Replica r of a single-process multi-device model lives on HIP device r mod device_count;
        the peers are created at the first fit, from the same seed as the primary."""
        if getattr(self, "_peers", None) is None:
            import copy

            from .engine import device_count, set_device

            if int(self.params.hp.device_rank) != 0 or self.params.counters() != (0, 0):
                raise RuntimeError("single-process multi-device fit needs the untrained rank-0 model")
            ndev = device_count()
            peers = []
            for r in range(1, int(self.params.hp.num_devices)):
                hp = copy.copy(self.params.hp)
                hp.device_rank = r
                set_device(r % ndev)
                peers.append(Model(hp))
            set_device(0)
            self._peers = peers
        return [self.params] + self._peers

    def user_representation(self, item_ids) -> ImplicitUser:
        return ImplicitUser(self.params.user_representation(np.asarray(item_ids, dtype=np.uint32)))

    def predict(self, user: ImplicitUser, item_ids) -> np.ndarray:
        return self.params.predict(user.user_embedding, np.asarray(item_ids, dtype=np.uint32))


class ImplicitLSTMModel(_ImplicitSequenceModel):
    """An LSTM-based sequence model for implicit feedback (lstm.rs:386-416)."""
