#!/usr/bin/env python3
"""Throughput of look-ahead steps grouped ACROSS games by the card played (design probe for a binned look-ahead kernel).

Every candidate (mid-game state, legal action) of BASE heuristic self-play games gets a lane of its own: engine 2 holds one
copy of the parent state per candidate (same seed + same weights -> the same trajectory), and k_step executes all
candidates at once, API_LANES lanes per wavefront (MONSOON_LIB selects a build with 16 / 32 / 64).  Orders:
  game    candidates of a game are neighbours (what a k_play pass sees)
  card    sorted by (card played, tile, game)
  random
The k_step durations are read from a rocprofv3 kernel trace of this script (scripts/probe/divergence2.sh).
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

BASE = int(os.environ.get("PROBE_BASE", "8192"))
ROUNDS = int(os.environ.get("PROBE_ROUNDS", "24"))
OFF_PL, PL_SIZE, P_HAND = 80, 96, 12


def main():
    rs = np.random.RandomState(7)
    deck = deck_indices("N12M")
    w = rs.uniform(0, 1, 10).reshape(1, 10)

    def midgame(seeds):
        e = BatchEngine(len(seeds))
        e.reset(np.asarray(seeds, dtype=np.uint32), np.stack([deck, deck]))
        e.upload_weights(w)
        e.assign_players(np.zeros(len(seeds), dtype=np.int32), np.zeros(len(seeds), dtype=np.int32))
        e.play_rounds(ROUNDS)
        e.sync()
        return e

    seeds = np.arange(BASE, dtype=np.uint32) + 5000
    e1 = midgame(seeds)
    masks = e1.legal_mask()
    h1 = e1.state_hash()
    cands = []   # (game, action, key)
    for g in range(BASE):
        acts = [a for a in range(156) if (int(masks[g][a >> 6]) >> (a & 63)) & 1]
        if len(acts) <= 1:
            continue
        raw = e1.debug_raw(g)
        lo = int(raw[0])
        for a in acts:
            if a < 148:
                ci = a >> 4 if a < 64 else (a - 64) // 21
                key = (int(raw[OFF_PL + lo * PL_SIZE + P_HAND + 4 * ci]), a & 15 if a < 64 else (a - 64) % 21)
            else:
                key = (200 + (a == 155), a)
            cands.append((g, a, key))
    n = len(cands) // 64 * 64
    cands = cands[:n]
    print(f"{n} candidates of {BASE} games ({n / BASE:.1f} per game), lib {os.environ.get('MONSOON_LIB', 'default')}", flush=True)
    orders = {"game": list(range(n)), "card": sorted(range(n), key=lambda i: (cands[i][2], cands[i][0])), "random": list(rs.permutation(n))}
    for tag in os.environ.get("PROBE_ORDERS", "game,card,random,card").split(","):
        o = orders[tag]
        e2 = midgame([seeds[cands[i][0]] for i in o])
        h2 = e2.state_hash()
        assert all(h2[s] == h1[cands[i][0]] for s, i in enumerate(o[:2000])), "copies differ from their base game"
        acts = np.array([cands[i][1] for i in o], dtype=np.uint8)
        t0 = time.time()
        r, d, f = e2.step(acts)
        dt = time.time() - t0
        print(f"order {tag}: {n} lanes stepped in {1e3 * dt:.1f} ms host wall (with PCIe + legality pre-pass), faults {int((f != 0).sum())}", flush=True)
        e2.close()


if __name__ == "__main__":
    main()
