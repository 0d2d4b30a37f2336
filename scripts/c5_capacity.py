#!/usr/bin/env python3
"""Capacity study for the extended record on C5 games (random 12-card decks of the 109 observable cards), CPU only.

    python scripts/c5_capacity.py [--games 32768] [--rem 16 --world 8]
Builds oracle/_cap/liboracle_<rem>_<world>.so (the host build of the rules core with those capacities), plays the games
and prints the histogram of monsoon_game_faults-style codes (the fault that stopped a game, else the first capacity code
one of its look-aheads hit).  Uses oracle/: test infrastructure, not a product path.
"""
import argparse
import collections
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib  # noqa: E402
from monsoon_amd.cards import CARD_IDS  # noqa: E402

W0 = np.random.RandomState(2024).uniform(0, 1, 10)   # the weight vector of tests/test_gpu_parity.py


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=32768)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--rem", type=int, default=16)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ext", type=int, default=1)
    ap.add_argument("--ent", type=int, default=128)
    ap.add_argument("--depth", type=int, default=40)
    ap.add_argument("--remdepth", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    d = os.path.join(REPO, "oracle", "_cap")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"liboracle_{args.ext}_{args.ent}_{args.rem}_{args.world}.so")
    flags = "-O2 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -fno-fast-math -Wno-psabi".split() + ["-I" + os.path.join(REPO, "monsoon_amd", "csrc")]
    subprocess.run(["g++", *flags, f"-DMSB_EXT={args.ext}", f"-DMSB_CAP_REM={args.rem}", f"-DMSB_CAP_WORLD={args.world}", f"-DMSB_CAP_ENT={args.ent}", f"-DMSB_CAP_DEPTH={args.depth}", f"-DMSB_CAP_REMDEPTH={args.remdepth}", "-shared", "-o", path,
                    os.path.join(REPO, "oracle", "oracle.cpp"), "-lpthread"], check=True)
    pool = np.array([i for i, c in enumerate(CARD_IDS) if c not in ("up01", "up02", "up03")], dtype=np.uint8)
    n = args.games
    orc = oracle_lib.Oracle(n, extended=path)
    orc.L.orc_game_fault.argtypes = [oracle_lib.ctypes.c_void_p, oracle_lib.ctypes.c_int]
    for g in range(n):
        k = args.first + g
        rs = np.random.RandomState(k ^ 0x9E3779B9)
        orc.reset(g, 90000 + k, rs.choice(pool, 12, replace=False), rs.choice(pool, 12, replace=False))
    t0 = time.time()
    total, res, steps, hashes = orc.rollout_batch(n, W0, 200, args.threads)
    dt = time.time() - t0
    codes = collections.Counter(orc.L.orc_game_fault(orc.h, g) for g in range(n))
    print(f"ent {args.ent} rem {args.rem} world {args.world}: {n} games, {total / 1e6:.1f} M look-aheads in {dt:.1f} s; fault codes {dict(sorted(codes.items()))}; "
          f"capacity {sum(v for k, v in codes.items() if k >= 16)} ({100 * sum(v for k, v in codes.items() if k >= 16) / n:.3f} %)")
    np.save(os.path.join(d, f"res_{args.ext}_{args.ent}_{args.rem}_{args.world}.npy"), np.stack([res.astype(np.int64), steps.astype(np.int64), hashes.astype(np.int64),
            np.array([orc.L.orc_game_fault(orc.h, g) for g in range(n)], dtype=np.int64)]))


if __name__ == "__main__":
    main()
