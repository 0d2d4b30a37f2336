#!/usr/bin/env python3
"""Where a rollout batch spends its time: reset (stream seeding + game construction) vs decision rounds.
    gpurun -- python scripts/rollout_timing.py [games]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
deck_name = sys.argv[2] if len(sys.argv) > 2 else "N12M"   # a named deck, or "random" = C5's per-game 12-card decks (extended build)
random_decks = deck_name == "random"
eng = BatchEngine(65536, extended=random_decks, stack_bytes=32768 if random_decks else 0)
W = np.random.RandomState(2024).uniform(0, 1, (2, 10))
if random_decks:
    from monsoon_amd.cards import CARD_IDS
    pool = np.array([i for i, c in enumerate(CARD_IDS) if c not in ("up01", "up02", "up03")], dtype=np.uint8)   # the 109 observable cards
    pairs = np.zeros((n, 2, 12), dtype=np.uint8)
    for g in range(n):
        rs = np.random.RandomState(g ^ 0x9E3779B9)
        pairs[g, 0] = rs.choice(pool, 12, replace=False)
        pairs[g, 1] = rs.choice(pool, 12, replace=False)
else:
    deck = deck_indices(deck_name)
    pairs = np.stack([deck, deck])[None]
for rep in range(3):
    t0 = time.perf_counter()
    eng.reset(np.arange(n, dtype=np.uint32) + rep * n, pairs if random_decks else pairs[0])
    t1 = time.perf_counter()
    matches = np.zeros(n, dtype=[("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])
    matches["p2"] = 1
    matches["seed"] = np.arange(n) + rep * n
    if random_decks:
        matches["deck"] = np.arange(n)
    eng.reset_stats()
    t2 = time.perf_counter()
    counts, res, steps = eng.rollout(W, matches, pairs, 200, want_results=True)
    t3 = time.perf_counter()
    ms, launches = eng.kernel_time()
    st = eng.stats()
    print(f"n={n} reset {1e3 * (t1 - t0):.1f} ms | rollout {1e3 * (t3 - t2):.1f} ms: k_play {ms:.1f} ms in {launches} launches, "
          f"{st['lookahead_steps'] / 1e6:.1f} M env-steps, mean game length {steps.mean():.1f}, "
          f"{st['lookahead_steps'] / (t3 - t2) / 1e6:.0f} M env-steps/s, faults {st['faults']} (build limits {st['capacity_faults']}), "
          f"results P1/P2/draw {int((res == 0).sum())}/{int((res == 1).sum())}/{int((res == -1).sum())}", flush=True)
    codes, cnt = np.unique(eng.game_faults(), return_counts=True)
    print("   fault codes:", {int(c): int(k) for c, k in zip(codes, cnt) if c}, flush=True)
