"""Differential check of the two implementations of the rules: the recursive oracle (oracle/recursive/) against the host
build of the product's work-stack core (monsoon_amd/csrc/rules.h), step by step.

    python scripts/core_diff.py [--games N] [--pool all|neutral|swarm] [--tier 0|1|2] [--heuristic] [--seed S]

Random-policy games (or heuristic self-play: every look-ahead of both cores is compared through the score vector) on
random decks; after every step: fault code, reward/done, legal mask and the canonical record must agree.  A step that
faults ends the game on both sides (the state behind a fault is not defined: canon.h).  CPU only, test infrastructure."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib  # noqa: E402
from monsoon_amd import cards  # noqa: E402


def pool_ids(name, tier):
    ids = list(range(112))
    names = cards.CARD_IDS
    if name == "neutral":
        ids = [i for i in ids if cards.deck_indices("N12M") is not None and names[i][0] in "ubs" and names[i][1] == "0"]
    elif name == "swarm":
        ids = list(cards.deck_indices("S12"))
    ext_only = {names.index("ua20"), names.index("b005")}
    if tier == 0:
        ids = [i for i in ids if i not in ext_only]
    return [i for i in ids if names[i] not in ("up01", "up02", "up03")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=200)
    ap.add_argument("--pool", default="all")
    ap.add_argument("--tier", type=int, default=0)
    ap.add_argument("--heuristic", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--must", default="")
    ap.add_argument("--max-steps", type=int, default=400)
    ap.add_argument("--core", default="product", help="product | product_evict (host build with the device's 21-word resident stack)")
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    ids = pool_ids(a.pool, a.tier)
    must = [cards.CARD_IDS.index(x) for x in a.must.split(",") if x]
    A = oracle_lib.Oracle(1, extended=a.tier, core="oracle")
    B = oracle_lib.Oracle(1, extended=a.tier, core=a.core)
    w = np.random.RandomState(2024).uniform(0, 1, 10)
    steps = faults = 0
    t0 = time.time()
    for g in range(a.games):
        seed = int(rs.randint(0, 2**31 - 1))
        decks = []
        for _ in range(2):
            rest = [i for i in ids if i not in must]
            d = list(must) + list(rs.choice(rest, 12 - len(must), replace=False))
            decks.append(np.array(d, dtype=np.uint8))
        fa = A.reset(0, seed, decks[0], decks[1])
        fb = B.reset(0, seed, decks[0], decks[1])
        assert fa == fb, ("reset fault", g, fa, fb)
        if fa:
            continue
        for t in range(a.max_steps):
            ma, mb = A.legal_mask(0), B.legal_mask(0)
            assert np.array_equal(ma, mb), ("legal", g, t)
            if a.heuristic:
                act, sa, _ = A.decide(0, w)
                actb, sb, _ = B.decide(0, w)
                ok = np.array_equal(np.nan_to_num(sa, nan=-7e77), np.nan_to_num(sb, nan=-7e77))
                assert ok and act == actb, ("decide", g, t, seed, [cards.CARD_IDS[i] for i in decks[0]], [cards.CARD_IDS[i] for i in decks[1]], act, actb,
                                           [(i, sa[i], sb[i]) for i in range(156) if not (sa[i] == sb[i] or (np.isnan(sa[i]) and np.isnan(sb[i])))])
                lfa, lfb = A.lookahead_faults(0), B.lookahead_faults(0)
                assert np.array_equal(lfa, lfb), ("lookahead faults", g, t, seed, lfa[lfa != lfb], lfb[lfa != lfb])
            else:
                legal = A.legal_actions(0)
                act = legal[rs.randint(len(legal))]
            ra, rb = A.step(0, act), B.step(0, act)
            steps += 1
            assert ra[0] == rb[0], ("fault", g, t, seed, act, ra, rb, [cards.CARD_IDS[i] for i in decks[0]], [cards.CARD_IDS[i] for i in decks[1]])
            if ra[0]:
                faults += 1
                break
            assert ra == rb, ("reward/done", g, t, ra, rb)
            ca, cb = A.canon(0), B.canon(0)
            if ca != cb:
                raise AssertionError(("canon", g, t, seed, act, [cards.CARD_IDS[i] for i in decks[0]], [cards.CARD_IDS[i] for i in decks[1]]))
            if A.have_winner(0):
                break
    print(f"{a.games} games, {steps} steps, {faults} ended by a fault, tier {a.tier}, pool {a.pool}: identical ({time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
