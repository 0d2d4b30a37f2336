#!/usr/bin/env python3
"""Per-call cost of small launches while other handles are alive (probe: the GPU test-suite got slow as its engine cache grew)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

deck = deck_indices("N12M")
import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
hip.hipDeviceSetLimit.argtypes = [ctypes.c_int, ctypes.c_size_t]
FIX = os.environ.get("FIX_LIMIT", "0") == "1"


def timed(eng, tag, k=50):
    if FIX:
        t0 = time.time()
        rc = hip.hipDeviceSetLimit(0, 32768 if eng.extended else 16384)   # hipLimitStackSize
        print(f"   hipDeviceSetLimit rc {rc} took {1e3 * (time.time() - t0):.2f} ms", flush=True)
    eng.reset(np.arange(eng.max_games, dtype=np.uint32), np.stack([deck, deck]))
    w = np.random.RandomState(1).uniform(0, 1, 10)
    eng.decide(w)
    t0 = time.time()
    for _ in range(k):
        eng.decide(w)
        eng.state_hash()
    print(f"{tag}: {1e3 * (time.time() - t0) / k:.2f} ms per decide+hash", flush=True)


a = BatchEngine(32)
timed(a, "std alone")
b = BatchEngine(32, extended=True)
timed(b, "ext, std alive")
timed(a, "std, ext alive")
c = BatchEngine(32, extended=2)
timed(c, "big, std+ext alive")
timed(a, "std, ext+big alive")
timed(b, "ext, std+big alive")
others = [BatchEngine(2048, lanes_per_game=u) for u in (4, 8, 16, 32, 64)] + [BatchEngine(65536)]
for e in others:
    e.reset(np.arange(e.max_games, dtype=np.uint32), np.stack([deck, deck]))
    e.decide(np.random.RandomState(1).uniform(0, 1, 10))
timed(a, "std, many alive")
timed(b, "ext, many alive")
timed(c, "big, many alive")
for e in others:
    e.close()
timed(c, "big, others closed")
b.close()
a.close()
timed(c, "big alone")
