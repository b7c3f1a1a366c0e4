"""ctypes view of include/sbr_hip.h (struct layout, enums, constants).  No library is loaded here."""
from __future__ import annotations

import ctypes as C
import enum


class Status(enum.IntEnum):
    OK = 0
    NO_INTERACTIONS = 1
    INVALID_PREDICTION = 2
    INVALID_ARGUMENT = 3
    UNSUPPORTED = 4
    NO_DEVICE = 5
    HIP = 6
    OUT_OF_MEMORY = 7


class ModelKind(enum.IntEnum):
    LSTM_NORMAL = 0
    LSTM_COUPLED = 1
    EWMA = 2


class Param(enum.IntEnum):
    ITEM_EMBEDDING = 0
    ITEM_EMBEDDING_ACC = 1
    ITEM_BIAS = 2
    ITEM_BIAS_ACC = 3
    LSTM_W = 4
    LSTM_W_ACC = 5
    LSTM_B = 6
    LSTM_B_ACC = 7
    EWMA_ALPHA = 8
    EWMA_ALPHA_ACC = 9
    ITEM_EMBEDDING_M = 10
    ITEM_BIAS_M = 11
    LSTM_W_M = 12
    LSTM_B_M = 13
    EWMA_ALPHA_M = 14


class Debug(enum.IntEnum):
    HIDDEN = 0
    NEGATIVES = 1
    COEF = 2
    LOSS = 3
    DHIDDEN = 4
    DINPUT = 5
    DENSE_GRAD = 6
    IN_IDX = 7
    OUT_IDX = 8
    TRIES = 9
    DZ = 10


class KernelFamily(enum.IntEnum):
    RECURRENT_FWD = 0
    SCORE = 1
    RECURRENT_BWD = 2
    DENSE_GRAD = 3
    DENSE_UPDATE = 4
    SPARSE_UPDATE = 5
    RANK = 6
    SPARSE_SORT = 7


NUM_KERNEL_FAMILIES = 8
ABI_VERSION = 10


class SbrHparams(C.Structure):
    _fields_ = [
        ("num_items", C.c_uint32),
        ("max_sequence_length", C.c_uint32),
        ("embedding_dim", C.c_uint32),
        ("learning_rate", C.c_float),
        ("l2_penalty", C.c_float),
        ("model", C.c_int32),
        ("loss", C.c_int32),
        ("optimizer", C.c_int32),
        ("parallelism", C.c_int32),
        ("seed", C.c_uint8 * 16),
        ("num_epochs", C.c_uint32),
        ("num_devices", C.c_uint32),
        ("device_rank", C.c_uint32),
        ("batch_sequences", C.c_uint32),
    ]


def storage_dim(embedding_dim: int) -> int:
    """Width the engine stores an embedding_dim in (16 / 32 / 64 / 128 / 256; the extra columns are zero and stay
    zero — sbr_engine.hip `storage_dim`).  Parameters, user representations and predictions use embedding_dim;
    the debug views and exchange blocks (`sbr_fit_debug_fetch`, dense gradient) use this width."""
    for p in (16, 32, 64, 128, 256):
        if 1 <= int(embedding_dim) <= p:
            return p
    raise ValueError(f"embedding_dim {embedding_dim}: 1..256 supported")


def make_hparams(num_items, max_sequence_length, embedding_dim, learning_rate, l2_penalty, model, loss,
                 optimizer, parallelism, seed, num_epochs, num_devices=1, device_rank=0,
                 batch_sequences=1) -> SbrHparams:
    hp = SbrHparams()
    hp.num_items = int(num_items)
    hp.max_sequence_length = int(max_sequence_length)
    hp.embedding_dim = int(embedding_dim)
    hp.learning_rate = float(learning_rate)
    hp.l2_penalty = float(l2_penalty)
    hp.model = int(model)
    hp.loss = int(loss)
    hp.optimizer = int(optimizer)
    hp.parallelism = int(parallelism)
    seed = bytes(seed)
    assert len(seed) == 16
    for i in range(16):
        hp.seed[i] = seed[i]
    hp.num_epochs = int(num_epochs)
    hp.num_devices = int(num_devices)
    hp.device_rank = int(device_rank)
    hp.batch_sequences = int(batch_sequences)
    return hp
