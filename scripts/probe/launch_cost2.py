#!/usr/bin/env python3
"""Which calls get slow when handles of several builds are alive: per-call times with the device stack limit printed."""
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
hip.hipDeviceGetLimit.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]
deck = deck_indices("N12M")
w = np.random.RandomState(1).uniform(0, 1, 10)


def limit():
    v = ctypes.c_size_t()
    hip.hipDeviceGetLimit(ctypes.byref(v), 0)
    return v.value


def t(fn, k=5):
    fn()
    t0 = time.time()
    for _ in range(k):
        fn()
    return 1e3 * (time.time() - t0) / k


def show(eng, tag):
    print(f"{tag:28s} limit {limit():6d}: decide {t(lambda: eng.decide(w)):8.2f} ms  hash {t(eng.state_hash):8.2f} ms  legal {t(eng.legal_mask):8.2f} ms", flush=True)


def mk(n, ext):
    e = BatchEngine(n, extended=ext)
    e.reset(np.arange(n, dtype=np.uint32), np.stack([deck, deck]))
    return e


a = mk(32, 0)
show(a, "std (only handle)")
b = mk(32, 1)
show(b, "ext (std alive)")
show(a, "std (ext alive)")
hip.hipDeviceSetLimit(0, 16384)
show(a, "std after SetLimit 16K")
show(b, "ext at 16K")
hip.hipDeviceSetLimit(0, 32768)
show(b, "ext after SetLimit 32K")
show(a, "std at 32K")
c = mk(32, 2)
show(c, "big (std+ext alive)")
show(c, "big again")
hip.hipDeviceSetLimit(0, 16384)
show(c, "big at 16K")
hip.hipDeviceSetLimit(0, 32768)
show(c, "big after SetLimit 32K")
show(c, "big again")
show(b, "ext")
show(a, "std at 32K")
big2 = mk(65536, 0)
show(big2, "std 65536 games at 16K?")
show(a, "std")
