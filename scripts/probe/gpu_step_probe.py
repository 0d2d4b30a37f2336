#!/usr/bin/env python3
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib
from monsoon_amd.cards import deck_indices
from monsoon_amd.engine import BatchEngine
deck = deck_indices("N12V")
n = 4
eng = BatchEngine(n)
eng.reset(np.arange(n, dtype=np.uint32), np.stack([deck, deck]))
orc = oracle_lib.Oracle(n)
for i in range(n):
    orc.reset(i, i, deck, deck)
for t in range(2):
    before = [eng.export(i) for i in range(n)]
    masks = eng.legal_mask()
    acts = np.array([eng.legal_actions(mask=masks[i])[0] for i in range(n)], dtype=np.uint8)
    r, d, f = eng.step(acts)
    print("t", t, "acts", acts, "reward", r, "done", d, "fault", f, flush=True)
    for i in range(n):
        fo, ro, do = orc.step(i, acts[i])
        after = eng.export(i)
        oc = orc.canon(i)
        diff = [k for k in range(min(len(after), len(oc))) if after[k] != oc[k]]
        print("  game", i, "oracle", (fo, ro, do), "equal", after == oc, "unchanged", after == before[i], "len", len(after), len(oc), "diff@", diff[:12], flush=True)
