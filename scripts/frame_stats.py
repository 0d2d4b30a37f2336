#!/usr/bin/env python3
"""How the work stack of the product's rules core is used (CPU, test infrastructure): handler invocations per frame type and
the depth of the stack, over heuristic self-play games on a study build of the host library.

    g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I monsoon_amd/csrc -DORC_PRODUCT_CORE -DMSB_COUNT_FRAMES \
        -shared -o oracle/_cap/libcount.so oracle/oracle.cpp -lpthread && python scripts/frame_stats.py [deck] [games]
Round 3, N12M, 64 games (282 154 steps): 3.08 handler invocations per step (F_MOVE 0.75, F_RUNAB 0.37, F_EACH 0.013, F_TURN 0.012,
F_DESTROY_TAIL 0.008; before F_STEP / F_UNIT_PLAY were removed they were 1.0 and 0.92), 98 % of them with at most 8 words on the
stack, deepest 22."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import oracle_lib  # noqa: E402
from monsoon_amd.cards import deck_indices  # noqa: E402

LIB = os.path.join(REPO, "oracle", "_cap", "libcount.so")
L = oracle_lib.lib(LIB, core="count")
L.orc_frame_counts.restype = ctypes.POINTER(ctypes.c_longlong)
c = L.orc_frame_counts()
o = oracle_lib.Oracle(1, extended=LIB, core="count")
deck = deck_indices(sys.argv[1] if len(sys.argv) > 1 else "N12M")
w = np.random.RandomState(2024).uniform(0, 1, 10)
steps = 0
for g in range(int(sys.argv[2]) if len(sys.argv) > 2 else 64):
    o.reset(0, g, deck, deck)
    r = o.rollout(0, w, w, 200)
    steps += r["lookahead"] + r["steps"]
names = ["all", "F_MOVE", "F_RUNAB", "F_CTXLEAVE", "F_DESTROY_TAIL", "F_CMD_TAIL", "F_EACH", "F_AFTER", "F_TURN", "F_EVICTED"]
print("steps (look-ahead + committed):", steps)
for i, n in enumerate(names):
    print(f"{n:16s} {c[i]:10d}  {c[i] / steps:6.3f} per step")
mv = ["ENTRY", "POISONED", "BEFORE_MOVING", "STEP", "BASE_HIT", "FIGHT", "STRUCK", "STRUCK_BACK", "TARGET_DEAD", "SELF_DEAD", "AFTER_ATTACK", "DONE", "START"]
print("F_MOVE by state:", {n: c[32 + i] for i, n in enumerate(mv) if c[32 + i]})
print("F_RUNAB by state:", {"start": c[48], "returned": c[49]})
print("inside move(): base hits", c[50], "fights", c[51], "advances tried", c[52])
print("deepest stack (words):", c[15])
print("handler invocations by stack depth / 4:", [c[16 + i] for i in range(12)])
