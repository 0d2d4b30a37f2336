#!/usr/bin/env python3
"""Minimal GPU probe: reset 8 games and compare the exported records with the oracle (exits non-zero on
any difference before touching anything else)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib
from monsoon_amd.cards import deck_indices
from monsoon_amd.engine import BatchEngine
deck = deck_indices("N12M")
eng = BatchEngine(8)
eng.reset(np.arange(8, dtype=np.uint32), np.stack([deck, deck]))
orc = oracle_lib.Oracle(8)
bad = 0
for i in range(8):
    orc.reset(i, i, deck, deck)
    bad += eng.export(i) != orc.canon(i)
print("reset mismatches:", bad, flush=True)
sys.exit(1 if bad else 0)
