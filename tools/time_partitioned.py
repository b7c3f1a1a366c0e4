#!/usr/bin/env python
"""Times the shape of BASELINE configs[4] (EWMA + hinge, d = 256, item table partitioned over the
replicas of one process) on whatever devices are present.  On a one-GPU box all replicas share the
device, so this measures the owner-computes protocol's overhead against the replicated group, not xGMI.

    python tools/time_partitioned.py [--items 2000000] [--users 200000] [--replicas 4] [--batch 16384]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=2_000_000)
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--replicas", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--only", choices=["both", "partitioned", "replicated"], default="both")
    args = ap.parse_args()
    from bench import synthetic_csr
    from sbr_rs_amd._abi import make_hparams
    from sbr_rs_amd.engine import device_count, group_create, group_fit

    ptr, items = synthetic_csr(args.users, args.items, args.max_len)
    nnz = int(ptr[-1]) - args.users
    print(f"devices {device_count()}, replicas {args.replicas}, {args.users} users x {args.items} items, d {args.dim}, "
          f"EWMA + hinge, {nnz} interactions per epoch")
    for name, part in (("replicated", False), ("partitioned", True)):
        if args.only not in ("both", name):
            continue
        hp = make_hparams(args.items, args.max_len, args.dim, 0.16, 0.0004, 2, 1, 0, 1, bytes([42] * 16), args.epochs,
                          args.replicas, 0, args.batch)
        t0 = time.perf_counter()
        models = group_create(hp, args.replicas, partition_item_table=part)
        t1 = time.perf_counter()
        loss = group_fit(models, ptr, items)   # includes the first epoch's packing
        t2 = time.perf_counter()
        loss = group_fit(models, ptr, items)
        t3 = time.perf_counter()
        print(f"{name:12s} create {t1 - t0:6.1f} s   fit#1 {t2 - t1:6.2f} s   fit#2 {t3 - t2:6.2f} s "
              f"({args.epochs * nnz / (t3 - t2) / 1e6:7.1f} M interactions/s)   loss {loss:.6f}")
        for m in models:
            m.close()


if __name__ == "__main__":
    main()
