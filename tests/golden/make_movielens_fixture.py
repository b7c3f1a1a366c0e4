"""Generates tests/golden/movielens_100k.npz from the reference's own test data file.

The reference's end-to-end tests (src/models/lstm.rs:450-520, src/models/ewma.rs:463-507) run on
MovieLens-100K, which they download from the crate's repository (src/datasets.rs:66-71); the
identical file is mounted at /root/reference/data.csv (header user_id,item_id,rating,timestamp,
100 000 rows).  The fixture keeps the three columns the models read (datasets.rs deserialises
into data::Interaction{user_id,item_id,timestamp}), in the CSV's row order — the order matters
because CompressedInteractions uses a *stable* sort and 75 772 rows sit in timestamp ties.

Run (in the authoring container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_movielens_fixture.py
"""
import csv
import os

import numpy as np

SRC = "/root/reference/data.csv"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "movielens_100k.npz")

users, items, ts = [], [], []
with open(SRC, newline="") as f:
    for row in csv.DictReader(f):
        users.append(int(row["user_id"]))
        items.append(int(row["item_id"]))
        ts.append(int(row["timestamp"]))
users = np.asarray(users, dtype=np.uint16)
items = np.asarray(items, dtype=np.uint16)
ts = np.asarray(ts, dtype=np.uint32)
assert users.shape[0] == 100000 and users.max() == 943 and items.max() == 1682
np.savez_compressed(DST, user_id=users, item_id=items, timestamp=ts)
print(DST, os.path.getsize(DST), "bytes")
