#!/usr/bin/env python3
"""One-GPU slice of C5 at scale: N random-deck games (extended build) against the CPU replay (test infrastructure:
uses oracle/ through tests/oracle_lib.py).  gpurun -- python scripts/c5_parity.py [N]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib  # noqa: E402
from monsoon_amd.cards import CARD_IDS  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
W0 = np.random.RandomState(2024).uniform(0, 1, 10)
pool = np.array([i for i, c in enumerate(CARD_IDS) if c not in ("up01", "up02", "up03")], dtype=np.uint8)
pairs = np.zeros((n, 2, 12), dtype=np.uint8)
for g in range(n):
    rs = np.random.RandomState(g ^ 0x9E3779B9)
    pairs[g, 0], pairs[g, 1] = rs.choice(pool, 12, replace=False), rs.choice(pool, 12, replace=False)
m = np.zeros(n, dtype=[("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])
m["seed"] = 500000 + np.arange(n)
m["deck"] = np.arange(n)
eng = BatchEngine(n, extended=True)
t0 = time.time()
_, results, steps = eng.rollout(W0[None], m, pairs, 200, want_results=True)
t1 = time.time()
hashes, faults = eng.state_hash(), eng.game_faults()
orc = oracle_lib.Oracle(n, extended=True)
for g in range(n):
    assert orc.reset(g, int(m["seed"][g]), pairs[g, 0], pairs[g, 1]) == 0
t2 = time.time()
total, ores, osteps, ohash = orc.rollout_batch(n, W0, 200, 16)
t3 = time.time()
bad = int((results != ores).sum() + (steps != osteps).sum() + (hashes != ohash).sum())
codes, cnt = np.unique(faults, return_counts=True)
print(f"{n} random-deck games: GPU rollout {t1 - t0:.2f} s, CPU replay (16 threads) {t3 - t2:.2f} s, {total / 1e6:.1f} M env-steps, "
      f"mismatching games {bad}, fault codes {dict(zip(codes.tolist(), cnt.tolist()))}")
# the replay tier of monsoon_amd/fitness.py: games the extended record cannot hold, again on the large record -- and the
# same on the CPU replay
from monsoon_amd.fitness import replace_capacity_faulted  # noqa: E402
from oracle_rollout import oracle_rollout_tier  # noqa: E402
counts = np.zeros((1, 3), dtype=np.int64)
counts[0] = [(results == 0).sum(), (results == -1).sum(), n]
rf = eng.rollout_faults(n)
assert np.array_equal(rf, faults)
big = BatchEngine(2048, extended=2)


def replay_hip(sub):
    c, r, s = big.rollout(W0[None], sub, pairs, 200, want_results=True)
    return c.astype(np.int64), r, s, big.rollout_faults(len(sub))


t4 = time.time()
k = replace_capacity_faulted(counts, results, steps, rf, m, replay_hip)
t5 = time.time()
ocounts = np.zeros((1, 3), dtype=np.int64)
ocounts[0] = [(ores == 0).sum(), (ores == -1).sum(), n]
of = np.array([orc.game_fault(g) for g in range(n)], dtype=np.uint8)
replace_capacity_faulted(ocounts, ores, osteps, of, m, lambda sub: oracle_rollout_tier(W0[None], sub, pairs, 200, 2))
bad2 = int((results != ores).sum() + (steps != osteps).sum() + (rf != of).sum() + (counts != ocounts).sum())
codes, cnt = np.unique(rf, return_counts=True)
print(f"replay tier: {k} games replayed on the large record in {t5 - t4:.2f} s, mismatching games {bad2}, "
      f"fault codes after replay {dict(zip(codes.tolist(), cnt.tolist()))}")
sys.exit(1 if bad or bad2 else 0)
