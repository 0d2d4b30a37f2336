#!/usr/bin/env python3
"""Launch cost of k_play with handles of three builds alive, each on a stream of its own (the default since round 3)
or all on the device's default stream (MONSOON_OWN_STREAM=0: round 2's work-around for the scratch-stack hand-over).

    gpurun -- 'python scripts/probe/own_stream.py; MONSOON_OWN_STREAM=0 python scripts/probe/own_stream.py'
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

deck = deck_indices("N12M")
w = np.random.RandomState(1).uniform(0, 1, 10)


def t(fn, k=8):
    fn()
    t0 = time.time()
    for _ in range(k):
        fn()
    return 1e3 * (time.time() - t0) / k


def mk(n, ext):
    e = BatchEngine(n, extended=ext)
    e.reset(np.arange(n, dtype=np.uint32), np.stack([deck, deck]))
    return e


print("MONSOON_OWN_STREAM =", os.environ.get("MONSOON_OWN_STREAM", "(default)"))
a = mk(32, 0)
print(f"std alone                 decide {t(lambda: a.decide(w)):7.2f} ms", flush=True)
b = mk(32, 1)
print(f"ext (std alive)           decide {t(lambda: b.decide(w)):7.2f} ms", flush=True)
print(f"std (ext alive)           decide {t(lambda: a.decide(w)):7.2f} ms", flush=True)
c = mk(32, 2)
print(f"big (std + ext alive)     decide {t(lambda: c.decide(w)):7.2f} ms", flush=True)
for k in range(3):
    print(f"round-robin std/ext/big   decide {t(lambda: a.decide(w), 2):7.2f} / {t(lambda: b.decide(w), 2):7.2f} / {t(lambda: c.decide(w), 2):7.2f} ms", flush=True)
