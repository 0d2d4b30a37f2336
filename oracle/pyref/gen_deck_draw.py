#!/usr/bin/env python3
"""Known answers for monsoon_draw_decks (configuration C5's per-game decks), by numpy itself:

    RandomState(seed).choice(pool, 12, replace=False)   twice per seed, pool = the 109 observable card indices

-> tests/golden/deck_draw_kat.npz {seeds u32[n], pairs u8[n][2][12], pool u8[109]}.  numpy's legacy RandomState stream is
frozen (NEP 19), so the vectors do not depend on the numpy version.  Needs numpy and monsoon_amd/card_ids.json only.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from monsoon_amd.cards import draw_random_decks_numpy, observable_pool  # noqa: E402

rs = np.random.RandomState(20260105)
seeds = np.concatenate([np.array([0, 1, 2, 42, 123, 2**31 - 1, 2**31, 2**32 - 1], dtype=np.uint32),
                        np.arange(1000, 1200, dtype=np.uint32) ^ np.uint32(0x9E3779B9),
                        rs.randint(0, 2**32, size=1024 - 208, dtype=np.uint64).astype(np.uint32)])
pool = observable_pool()
pairs = draw_random_decks_numpy(seeds, pool)
out = os.path.join(REPO, "tests", "golden", "deck_draw_kat.npz")
np.savez_compressed(out, seeds=seeds, pairs=pairs, pool=pool)
print(out, len(seeds), "seeds")
