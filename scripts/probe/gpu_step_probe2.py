#!/usr/bin/env python3
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib
from monsoon_amd.cards import deck_indices
from monsoon_amd.engine import BatchEngine
deck = deck_indices("N12V")
n = 70
for skip0 in (False,):
    eng = BatchEngine(n)
    eng.reset(np.arange(n, dtype=np.uint32), np.stack([deck, deck]))
    orc = oracle_lib.Oracle(n)
    for i in range(n):
        orc.reset(i, i, deck, deck)
    masks = eng.legal_mask()
    acts = np.array([eng.legal_actions(mask=masks[i])[0] for i in range(n)], dtype=np.uint8)
    if skip0:
        acts[0] = 255
    r, d, f = eng.step(acts)
    bad = []
    for i in range(n):
        if acts[i] != 255:
            orc.step(i, acts[i])
        if eng.export(i) != orc.canon(i):
            bad.append(i)
    import ctypes
    for i in bad[:4] + [1]:
        raw = eng.debug_raw(i)
        oraw = np.zeros(len(raw), dtype=np.uint8)
        orc.L.orc_raw(orc.h, i, oraw.ctypes.data_as(ctypes.c_void_p))
        diff = [k for k in range(len(raw)) if raw[k] != oraw[k] and not (20 <= k < 24 or 64 <= k < 80)]
        print(" game", i, "raw diff offsets", diff[:40], "gpu", [int(raw[k]) for k in diff[:16]], "orc", [int(oraw[k]) for k in diff[:16]], flush=True)
    print("skip0", skip0, "bad games", bad, "fault nonzero", list(np.nonzero(f)[0]), flush=True)
    eng.close()
