#!/usr/bin/env python3
"""How well do look-ahead steps of DIFFERENT games share instructions inside one wavefront?

A design probe, not part of the product.  Takes 4 096 mid-game states (heuristic self-play, N12M decks), draws one random
legal card play per game and executes the same (state, action) pairs through k_step (one lane per game, 64 games per
wavefront) under four lane groupings.  Run under `rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
--kernel-trace` and compare the k_step dispatches (printed in order):

  A  games in random order (64 unrelated steps per wavefront)
  D  the same pairs sorted by (card played, tile)          -> what binning by card across games would give
  B  8 lanes = 8 consecutive legal actions of one game, 8 games per wavefront (the shape of a k_play pass x 8)
  C  all 64 lanes the same game and action                  -> no divergence at all
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

N = 4096
OFF_PL, PL_SIZE, P_HAND = 80, 96, 12


def legal_list(mask_row):
    return [a for a in range(156) if (int(mask_row[a >> 6]) >> (a & 63)) & 1]


def main():
    rs = np.random.RandomState(7)
    eng = BatchEngine(N)
    deck = deck_indices("N12M")
    eng.reset(np.arange(N, dtype=np.uint32) + 1000, np.stack([deck, deck]))
    eng.upload_weights(rs.uniform(0, 1, 10).reshape(1, 10))
    eng.assign_players(np.zeros(N, dtype=np.int32), np.zeros(N, dtype=np.int32))
    eng.play_rounds(int(os.environ.get("PROBE_ROUNDS", "24")))
    eng.sync()
    status = eng.status()
    masks = eng.legal_mask().reshape(N, 3)
    blobs, plays, keys, cons = [], [], [], []
    for g in range(N):
        acts = [a for a in legal_list(masks[g]) if a < 148]
        if not acts:
            continue
        raw = eng.debug_raw(g)
        lo = int(raw[0])
        a = acts[rs.randint(len(acts))]
        ci = a >> 4 if a < 64 else (a - 64) // 21
        card = int(raw[OFF_PL + lo * PL_SIZE + P_HAND + 4 * ci])
        blobs.append(eng.save_state(g))
        plays.append(a)
        keys.append((card, a & 15 if a < 64 else (a - 64) % 21))
        k = rs.randint(len(acts))
        cons.append([acts[(k + i) % len(acts)] for i in range(8)])
    m = len(blobs) // 64 * 64
    print(f"{m} playable mid-game states of {N}; distinct cards played: {len(set(k[0] for k in keys[:m]))}", flush=True)
    eng2 = BatchEngine(m)
    eng2.reset(np.arange(m, dtype=np.uint32), np.stack([deck, deck]))

    def run(tag, order, actions):
        for slot, g in enumerate(order):
            eng2.load_state(slot, blobs[g])
        eng2.sync()
        r, d, f = eng2.step(np.asarray(actions, dtype=np.uint8))
        print(f"{tag}: stepped {len(order)} lanes, faults {int((np.asarray(f) != 0).sum())}", flush=True)

    order_a = list(rs.permutation(m))
    run("A random", order_a, [plays[g] for g in order_a])
    order_d = sorted(range(m), key=lambda g: keys[g])
    run("D sorted by card", order_d, [plays[g] for g in order_d])
    order_b = [g for g in range(m // 8) for _ in range(8)]
    run("B 8 actions of a game", order_b, [cons[g][i] for g in range(m // 8) for i in range(8)])
    order_c = [g for g in range(m // 64) for _ in range(64)]
    run("C one game per wave", order_c, [plays[g] for g in order_c])


if __name__ == "__main__":
    main()
