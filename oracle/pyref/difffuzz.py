#!/usr/bin/env python3
"""Live differential check: Python reference vs the C++ restatement (liboracle.so).

TEST INFRASTRUCTURE (build container only; needs /root/reference).
For every (seed, deck pair) it plays a seeded random legal policy on the reference and mirrors
every action on the oracle, comparing after each step: the sorted legal-action list, reward,
done, the canonical state record byte for byte, the (27,5,4) observation and the 10 features.

Usage: PYTHONHASHSEED=0 python oracle/pyref/difffuzz.py --deck N12M --games 50 [--steps 300]
       ... --pool neutral   random 12-card decks drawn from a pool of supported cards
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness as H  # noqa: E402

from evo.features import StateFeatures  # noqa: E402

BUILD = int(os.environ.get("MSB_EXT", "0"))   # MSB_EXT=1: extended record (ua20, b005); MSB_EXT=2: the large record
EXT = BUILD >= 1
# MSB_CORE=product: the host build of the PRODUCT's rules core (explicit work stack) instead of the recursive oracle -- the
# same C entry points over the other implementation (oracle/Makefile)
CORE = os.environ.get("MSB_CORE", "oracle")
LIB = ctypes.CDLL(os.path.join(H.REPO, "oracle", {"oracle": ("liboracle.so", "liboracle_ext.so", "liboracle_big.so"),
                                                  "product": ("libproduct_host.so", "libproduct_host_ext.so", "libproduct_host_big.so")}[CORE][BUILD]))
LIB.orc_create.restype = ctypes.c_void_p
LIB.orc_canon_hash.restype = ctypes.c_uint64
for name in ("orc_reset", "orc_legal", "orc_step", "orc_observe", "orc_features", "orc_canon", "orc_destroy",
             "orc_canon_hash", "orc_have_winner", "orc_to_play", "orc_decide", "orc_expert_action"):
    getattr(LIB, name).argtypes = None

UNSUPPORTED = set() if EXT else {"ua20", "b005"}
FAULT_CARDS = {"up01", "up02", "up03"}


def deck_bytes(deck):
    return (ctypes.c_uint8 * 12)(*[H.CARD_INDEX[c] for c in deck])


class Mirror:
    def __init__(self):
        self.h = ctypes.c_void_p(LIB.orc_create(1))

    def reset(self, seed, d0, d1, f0=0, f1=0):
        return LIB.orc_reset(self.h, 0, ctypes.c_uint32(seed), deck_bytes(d0), deck_bytes(d1), f0, f1)

    def legal(self):
        m = (ctypes.c_uint64 * 3)()
        LIB.orc_legal(self.h, 0, m)
        return [a for a in range(156) if (m[a >> 6] >> (a & 63)) & 1]

    def step(self, a):
        r, d = ctypes.c_int(), ctypes.c_int()
        f = LIB.orc_step(self.h, 0, a, ctypes.byref(r), ctypes.byref(d))
        return f, r.value, d.value

    def canon(self):
        buf = (ctypes.c_uint8 * 2048)()
        n = LIB.orc_canon(self.h, 0, buf)
        return bytes(buf[:n])

    def observe(self):
        out = np.zeros(540, dtype=np.int32)
        r = LIB.orc_observe(self.h, 0, out.ctypes.data_as(ctypes.c_void_p))
        return None if r else out.reshape(27, 5, 4)

    def features(self):
        f = np.zeros(10)
        r = LIB.orc_features(self.h, 0, f.ctypes.data_as(ctypes.c_void_p))
        return None if r else f


def describe_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return f"first diff at byte {i}: ref={a[max(0,i-8):i+8].hex()} ours={b[max(0,i-8):i+8].hex()} (len {len(a)} vs {len(b)})"
    return f"length {len(a)} vs {len(b)}"


def play(seed, d0, d1, steps, policy_seed, verbose=False, expert=False):
    g = H.make_game(seed, d0, d1)
    mir = Mirror()
    f = mir.reset(seed, d0, d1)
    assert f == 0, f"reset fault {f}"
    pol = np.random.RandomState(policy_seed)
    ca, cb = H.canon(g), mir.canon()
    if ca != cb:
        return f"seed {seed}: initial state differs: {describe_diff(ca, cb)}", 0
    n = 0
    for t in range(steps):
        la = g.legal_actions()
        lb = mir.legal()
        if la != lb:
            return f"seed {seed} step {t}: legal differs ref={la} ours={lb}", n
        if expert:
            # both sides are the reference's scripted bot (games/stormbound.py:563-637); it draws from the game stream
            try:
                a = int(g.expert_action())
            except Exception as e:  # noqa: BLE001
                fo = ctypes.c_int()
                LIB.orc_expert_action(mir.h, 0, ctypes.byref(fo))
                if fo.value == 0:
                    return f"seed {seed} step {t}: reference expert raised {e!r}, ours did not", n
                return None, n
            fo = ctypes.c_int()
            b = LIB.orc_expert_action(mir.h, 0, ctypes.byref(fo))
            if fo.value != 0 or a != b:
                return f"seed {seed} step {t}: expert action ref={a} ours={b} fault={fo.value}", n
        else:
            a = int(la[pol.randint(0, len(la))])
        ref_exc = None
        try:
            obs, reward, done = g.step(a)
        except Exception as e:  # noqa: BLE001
            ref_exc = e
        fb, rb, db = mir.step(a)
        n += 1
        if ref_exc is not None:
            if fb == 0 and mir.observe() is not None:
                return f"seed {seed} step {t} action {a}: reference raised {ref_exc!r}, ours did not", n
            return None, n  # both faulted: the game is over for the agent layer
        if fb != 0:
            return f"seed {seed} step {t} action {a}: ours faulted ({fb}), reference did not", n
        ca, cb = H.canon(g), mir.canon()
        if ca != cb:
            return f"seed {seed} step {t} action {a}: state differs: {describe_diff(ca, cb)}", n
        if (reward, int(done)) != (rb, db):
            return f"seed {seed} step {t} action {a}: reward/done ref={(reward, done)} ours={(rb, db)}", n
        ob = mir.observe()
        if ob is None or not np.array_equal(obs, ob):
            bad = np.argwhere(obs != ob) if ob is not None else None
            return f"seed {seed} step {t} action {a}: observation differs at {bad[:5].tolist() if bad is not None else 'raise'}", n
        fa = StateFeatures(obs, g.to_play()).get_feature_vector()
        fbv = mir.features()
        if not np.array_equal(fa.view(np.uint64), fbv.view(np.uint64)):
            return f"seed {seed} step {t}: features differ ref={fa} ours={fbv}", n
        if g.have_winner():
            break
    return None, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--deck", default="N12M")
    ap.add_argument("--deck2", default=None)
    ap.add_argument("--pool", default=None, help="'all' = random decks from every supported card")
    ap.add_argument("--games", type=int, default=20)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--expert", action="store_true", help="drive both sides with Stormbound.expert_action")
    ap.add_argument("--must", default="", help="comma list of card ids forced into both random decks (with --pool)")
    args = ap.parse_args()
    total, bad = 0, 0
    pool = None
    if args.pool:
        pool = [c for c in H.CARD_IDS if c not in UNSUPPORTED and c not in FAULT_CARDS]
    for k in range(args.games):
        seed = args.seed0 + k
        if pool:
            rs = np.random.RandomState(seed ^ 0x9E3779B9)
            must = [c for c in args.must.split(",") if c]
            rest = [c for c in pool if c not in must]
            d0 = must + list(rs.choice(rest, 12 - len(must), replace=False))
            d1 = must + list(rs.choice(rest, 12 - len(must), replace=False))
        else:
            d0 = H.DECKS[args.deck]
            d1 = H.DECKS[args.deck2 or args.deck]
        err, n = play(seed, d0, d1, args.steps, policy_seed=seed + 1000, expert=args.expert)
        total += n
        if err:
            bad += 1
            print("MISMATCH", err)
            if pool:
                print("   decks", d0, d1)
        if (k + 1) % 10 == 0:
            print(f"progress: {k + 1} games, {total} steps, {bad} mismatching games", file=sys.stderr, flush=True)
    print(f"{args.games} games, {total} steps, {bad} mismatching games")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
