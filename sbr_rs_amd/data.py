"""Interaction containers — host-side mirror of the reference's ``sbr::data`` module
(/root/reference/src/data.rs).  Same names, same argument meaning, same results for the parts the
reference pins (chunking, CSR conversion, user-hash split); numpy instead of ``Vec``.

The CSR produced by :meth:`Interactions.to_compressed` (``user_pointers`` u64, ``item_ids`` u32,
time-sorted per user with a *stable* sort) is exactly the input format of the C-ABI
(``sbr_model_fit`` / ``sbr_mrr_score`` in include/sbr_hip.h).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .rng import XorShiftRng

_M64 = (1 << 64) - 1


@dataclass(frozen=True)
class Interaction:
    """Basic interaction type (data.rs:17-51)."""

    _user_id: int
    _item_id: int
    _timestamp: int

    def user_id(self) -> int:
        return self._user_id

    def item_id(self) -> int:
        return self._item_id

    def weight(self) -> float:
        return 1.0

    def timestamp(self) -> int:
        return self._timestamp


def _siphash24_u64(key0: int, key1: int, values: np.ndarray) -> np.ndarray:
    """SipHash-2-4 of each value written as 8 little-endian bytes (``Hasher::write_usize`` on a
    64-bit target, data.rs:81-84; siphasher 0.2 ``SipHasher`` = SipHash-2-4).  Vectorised."""
    v = values.astype(np.uint64)
    n = v.shape[0]
    with np.errstate(over="ignore"):
        k0 = np.uint64(key0)
        k1 = np.uint64(key1)
        v0 = np.full(n, k0 ^ np.uint64(0x736F6D6570736575), dtype=np.uint64)
        v1 = np.full(n, k1 ^ np.uint64(0x646F72616E646F6D), dtype=np.uint64)
        v2 = np.full(n, k0 ^ np.uint64(0x6C7967656E657261), dtype=np.uint64)
        v3 = np.full(n, k1 ^ np.uint64(0x7465646279746573), dtype=np.uint64)

        def rotl(x, b):
            return (x << np.uint64(b)) | (x >> np.uint64(64 - b))

        def sipround(v0, v1, v2, v3):
            v0 = v0 + v1
            v1 = rotl(v1, 13)
            v1 = v1 ^ v0
            v0 = rotl(v0, 32)
            v2 = v2 + v3
            v3 = rotl(v3, 16)
            v3 = v3 ^ v2
            v0 = v0 + v3
            v3 = rotl(v3, 21)
            v3 = v3 ^ v0
            v2 = v2 + v1
            v1 = rotl(v1, 17)
            v1 = v1 ^ v2
            v2 = rotl(v2, 32)
            return v0, v1, v2, v3

        # one full 8-byte message word
        m = v
        v3 = v3 ^ m
        v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
        v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
        v0 = v0 ^ m
        # final block: length (8) in the top byte, no tail bytes
        b = np.uint64(8 << 56)
        v3 = v3 ^ b
        v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
        v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
        v0 = v0 ^ b
        v2 = v2 ^ np.uint64(0xFF)
        for _ in range(4):
            v0, v1, v2, v3 = sipround(v0, v1, v2, v3)
        return v0 ^ v1 ^ v2 ^ v3


class Interactions:
    """A collection of individual interactions (data.rs:92-211)."""

    def __init__(self, num_users: int, num_items: int):
        self._num_users = int(num_users)
        self._num_items = int(num_items)
        self._users = np.zeros(0, dtype=np.uint64)
        self._items = np.zeros(0, dtype=np.uint64)
        self._timestamps = np.zeros(0, dtype=np.uint64)

    # -- construction --------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, users, items, timestamps, num_users: Optional[int] = None,
                    num_items: Optional[int] = None) -> "Interactions":
        """``impl From<Vec<Interaction>>`` (data.rs:200-211): num_users = max id + 1."""
        users = np.asarray(users, dtype=np.uint64)
        items = np.asarray(items, dtype=np.uint64)
        timestamps = np.asarray(timestamps, dtype=np.uint64)
        if num_users is None:
            num_users = int(users.max()) + 1
        if num_items is None:
            num_items = int(items.max()) + 1
        out = cls(num_users, num_items)
        out._users, out._items, out._timestamps = users.copy(), items.copy(), timestamps.copy()
        return out

    @classmethod
    def from_vec(cls, interactions: Sequence[Interaction]) -> "Interactions":
        return cls.from_arrays([x.user_id() for x in interactions], [x.item_id() for x in interactions],
                               [x.timestamp() for x in interactions])

    def push(self, interaction: Interaction) -> None:
        self._users = np.append(self._users, np.uint64(interaction.user_id()))
        self._items = np.append(self._items, np.uint64(interaction.item_id()))
        self._timestamps = np.append(self._timestamps, np.uint64(interaction.timestamp()))

    # -- accessors -----------------------------------------------------------------------------
    def data(self) -> List[Interaction]:
        return [Interaction(int(u), int(i), int(t)) for u, i, t in zip(self._users, self._items, self._timestamps)]

    def arrays(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        return self._users, self._items, self._timestamps

    def len(self) -> int:
        return int(self._users.shape[0])

    __len__ = len

    def is_empty(self) -> bool:
        return self.len() == 0

    def num_users(self) -> int:
        return self._num_users

    def num_items(self) -> int:
        return self._num_items

    def shape(self) -> Tuple[int, int]:
        return (self._num_users, self._num_items)

    # -- transformations -----------------------------------------------------------------------
    def _take(self, idx) -> "Interactions":
        out = Interactions(self._num_users, self._num_items)
        out._users, out._items, out._timestamps = self._users[idx], self._items[idx], self._timestamps[idx]
        return out

    def shuffle(self, rng: XorShiftRng) -> None:
        """In-place Fisher-Yates with the caller's RNG (data.rs:128-130)."""
        perm = rng.permutation(self.len())
        self._users, self._items, self._timestamps = self._users[perm], self._items[perm], self._timestamps[perm]

    def split_at(self, idx: int) -> Tuple["Interactions", "Interactions"]:
        return self._take(slice(0, idx)), self._take(slice(idx, None))

    def split_by(self, func: Callable[[Interaction], bool]) -> Tuple["Interactions", "Interactions"]:
        mask = np.array([bool(func(x)) for x in self.data()], dtype=bool) if self.len() else np.zeros(0, bool)
        return self._take(mask), self._take(~mask)

    def _split_by_mask(self, mask: np.ndarray) -> Tuple["Interactions", "Interactions"]:
        return self._take(mask), self._take(~mask)

    def to_compressed(self) -> "CompressedInteractions":
        return CompressedInteractions.from_interactions(self)

    def to_triplet(self) -> "TripletInteractions":
        """COO form (data.rs:132-134, 558-575): three parallel arrays in the interactions' order."""
        return TripletInteractions(self._num_users, self._num_items, self._users, self._items, self._timestamps)


def train_test_split(interactions: Interactions, rng: XorShiftRng, test_fraction: float):
    """Random split (data.rs:54-64): shuffle in place, the first ``test_fraction`` is the test set."""
    interactions.shuffle(rng)
    cut = int(np.float32(test_fraction) * np.float32(interactions.len()))
    test, train = interactions.split_at(cut)
    return train, test


def user_based_split(interactions: Interactions, rng: XorShiftRng, test_fraction: float):
    """Split so that no user is in both sets (data.rs:69-88): two u64 keys from ``rng``,
    SipHash-2-4 of the user id, train iff ``hash % 100000 > (test_fraction * 100000) as u64``."""
    denominator = 100_000
    train_cutoff = int(np.float32(test_fraction) * np.float32(denominator))
    key_0 = rng.uniform(0, _M64)  # Uniform::new(0, std::u64::MAX).sample(rng), data.rs:77-78
    key_1 = rng.uniform(0, _M64)
    users = interactions._users
    uniq, inverse = np.unique(users, return_inverse=True)
    hashes = _siphash24_u64(key_0, key_1, uniq)
    is_train = (hashes % np.uint64(denominator)) > np.uint64(train_cutoff)
    return interactions._split_by_mask(is_train[inverse] if users.shape[0] else np.zeros(0, bool))


@dataclass
class CompressedInteractionsUser:
    """A single user's data, earliest to latest (data.rs:339-371)."""

    user_id: int
    item_ids: np.ndarray
    timestamps: np.ndarray

    def len(self) -> int:
        return int(self.item_ids.shape[0])

    __len__ = len

    def is_empty(self) -> bool:
        return self.len() == 0

    def chunks(self, chunk_size: int) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        """Chunked iterator: the FIRST chunk is the smallest, the rest are ``chunk_size``
        (data.rs:363-370, 406-431)."""
        user_len = self.len()
        idx = 0
        while idx < user_len:
            mod = (user_len - idx) % chunk_size
            size = chunk_size if mod == 0 else mod
            yield self.item_ids[idx:idx + size], self.timestamps[idx:idx + size]
            idx += size


class CompressedInteractions:
    """CSR by user, time-sorted (data.rs:227-329)."""

    def __init__(self, num_users: int, num_items: int, user_pointers: np.ndarray, item_ids: np.ndarray,
                 timestamps: np.ndarray):
        self._num_users = int(num_users)
        self._num_items = int(num_items)
        self.user_pointers = np.ascontiguousarray(user_pointers, dtype=np.uint64)
        self.item_ids = np.ascontiguousarray(item_ids, dtype=np.uint32)
        self.timestamps = np.ascontiguousarray(timestamps, dtype=np.uint64)

    @classmethod
    def from_interactions(cls, interactions: Interactions) -> "CompressedInteractions":
        """data.rs:236-265 — ``sort_by(cmp_timestamp)`` is a stable sort on (user, timestamp), so
        timestamp ties keep their input order."""
        users, items, ts = interactions.arrays()
        order = np.lexsort((ts, users))  # lexsort is stable; last key is primary
        users, items, ts = users[order], items[order], ts[order]
        counts = np.bincount(users.astype(np.int64), minlength=interactions.num_users()) if users.shape[0] else \
            np.zeros(interactions.num_users(), dtype=np.int64)
        ptr = np.zeros(interactions.num_users() + 1, dtype=np.uint64)
        ptr[1:] = np.cumsum(counts).astype(np.uint64)
        return cls(interactions.num_users(), interactions.num_items(), ptr, items.astype(np.uint32), ts)

    def iter_users(self) -> Iterator[CompressedInteractionsUser]:
        for u in range(self._num_users):
            yield self.get_user(u)

    def get_user(self, user_id: int) -> Optional[CompressedInteractionsUser]:
        if user_id >= self._num_users:
            return None
        start, stop = int(self.user_pointers[user_id]), int(self.user_pointers[user_id + 1])
        return CompressedInteractionsUser(user_id, self.item_ids[start:stop], self.timestamps[start:stop])

    def num_users(self) -> int:
        return self._num_users

    def num_items(self) -> int:
        return self._num_items

    def shape(self) -> Tuple[int, int]:
        return (self._num_users, self._num_items)

    def to_interactions(self) -> Interactions:
        counts = np.diff(self.user_pointers.astype(np.int64))
        users = np.repeat(np.arange(self._num_users, dtype=np.uint64), counts)
        return Interactions.from_arrays(users, self.item_ids, self.timestamps, self._num_users, self._num_items)


class TripletMinibatch:
    """A minibatch of triplet interactions: views of the three arrays (data.rs:502-520)."""

    def __init__(self, user_ids: np.ndarray, item_ids: np.ndarray, timestamps: np.ndarray):
        self.user_ids, self.item_ids, self.timestamps = user_ids, item_ids, timestamps

    def len(self) -> int:
        return int(self.user_ids.shape[0])

    __len__ = len

    def is_empty(self) -> bool:
        return self.item_ids.shape[0] == 0


class TripletMinibatchIterator:
    """Minibatches of exactly ``minibatch_size`` interactions over [idx, stop_idx): a shorter remainder is never yielded
    (data.rs:522-545)."""

    def __init__(self, interactions: "TripletInteractions", idx: int, stop_idx: int, minibatch_size: int):
        self._interactions, self._idx, self._stop_idx, self._minibatch_size = interactions, int(idx), int(stop_idx), int(minibatch_size)

    def slice(self, start: int, stop: int) -> "TripletMinibatchIterator":
        """An iterator over [start, stop) of the same data with the same minibatch size (data.rs:491-499)."""
        return TripletMinibatchIterator(self._interactions, start, stop, self._minibatch_size)

    def __iter__(self) -> "TripletMinibatchIterator":
        return self

    def __next__(self) -> TripletMinibatch:
        start, stop = self._idx, self._idx + self._minibatch_size
        self._idx = stop
        if stop > self._stop_idx:
            raise StopIteration
        t = self._interactions
        return TripletMinibatch(t.user_ids[start:stop], t.item_ids[start:stop], t.timestamps[start:stop])


class TripletInteractions:
    """Interactions in COO form (data.rs:435-481).  Not consumed by the sequence models; kept for API parity."""

    def __init__(self, num_users: int, num_items: int, user_ids, item_ids, timestamps):
        self._num_users, self._num_items = int(num_users), int(num_items)
        self.user_ids = np.ascontiguousarray(user_ids, dtype=np.uint64)
        self.item_ids = np.ascontiguousarray(item_ids, dtype=np.uint64)
        self.timestamps = np.ascontiguousarray(timestamps, dtype=np.uint64)

    def len(self) -> int:
        return int(self.user_ids.shape[0])

    __len__ = len

    def is_empty(self) -> bool:
        return self.len() == 0

    def iter_minibatch(self, minibatch_size: int) -> TripletMinibatchIterator:
        return TripletMinibatchIterator(self, 0, self.len(), minibatch_size)

    def iter_minibatch_partitioned(self, minibatch_size: int, num_partitions: int) -> List[TripletMinibatchIterator]:
        """``num_partitions`` iterators over consecutive slices of len / num_partitions interactions (integer division: the
        remainder belongs to no partition, data.rs:463-475)."""
        iterator = self.iter_minibatch(minibatch_size)
        chunk_size = self.len() // num_partitions
        return [iterator.slice(x * chunk_size, (x + 1) * chunk_size) for x in range(num_partitions)]

    def num_users(self) -> int:
        return self._num_users

    def num_items(self) -> int:
        return self._num_items

    def shape(self) -> Tuple[int, int]:
        return (self._num_users, self._num_items)
