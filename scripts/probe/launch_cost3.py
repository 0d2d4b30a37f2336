#!/usr/bin/env python3
"""k_play launch latency against the device stack limit, per build (probe)."""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipDeviceSetLimit.argtypes = [ctypes.c_int, ctypes.c_size_t]
deck = deck_indices("N12M")
w = np.random.RandomState(1).uniform(0, 1, 10)


def t(fn, k=4):
    fn()
    t0 = time.time()
    for _ in range(k):
        fn()
    return 1e3 * (time.time() - t0) / k


for ext, n in ((0, 32), (1, 32), (2, 32), (0, 8192)):
    e = BatchEngine(n, extended=ext)
    e.reset(np.arange(n, dtype=np.uint32), np.stack([deck, deck]))
    row = []
    for kb in (8, 12, 16, 20, 24, 28, 32, 40, 48, 64):
        hip.hipDeviceSetLimit(0, kb * 1024)
        row.append(f"{kb}K {t(lambda: e.decide(w)):7.2f}")
    print(f"build {ext} n {n} variant {e.variant()}: " + " | ".join(row), flush=True)
    e.close()
