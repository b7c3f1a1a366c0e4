"""Dataset ingestion — mirror of the parts of `sbr::datasets` that do not need the network
(/root/reference/src/datasets.rs:36-71).  `download_movielens_100k` fetches `data.csv` over HTTP
in the reference; there is no egress here, so the loader takes a local path (CSV with the header
`user_id,item_id,rating,timestamp`, deserialised into `Interaction{user_id,item_id,timestamp}` as
at datasets.rs:57-60) or the compact `.npz` fixture tests/golden/movielens_100k.npz."""
from __future__ import annotations

import csv
import os

import numpy as np

from .data import Interactions


def load_csv(path: str) -> Interactions:
    users, items, ts = [], [], []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            users.append(int(row["user_id"]))
            items.append(int(row["item_id"]))
            ts.append(int(row["timestamp"]))
    return Interactions.from_arrays(users, items, ts)


def load_npz(path: str) -> Interactions:
    z = np.load(path)
    return Interactions.from_arrays(z["user_id"], z["item_id"], z["timestamp"])


def download_movielens_100k(path: str = None) -> Interactions:
    """Same name as the reference entry point; reads a local copy instead of downloading."""
    path = path or os.environ.get("SBR_MOVIELENS_PATH") or os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "movielens_100k.npz")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: no network in this environment — point SBR_MOVIELENS_PATH at data.csv or the .npz fixture")
    return load_npz(path) if path.endswith(".npz") else load_csv(path)
