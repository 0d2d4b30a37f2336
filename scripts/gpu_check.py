#!/usr/bin/env python3
"""First-light GPU check: HIP engine vs the CPU oracle on seeded N12M games (run via gpurun)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib  # noqa: E402
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

U = int(os.environ.get("MSB_U", "32"))
n = int(os.environ.get("MSB_N", "64"))
deck = deck_indices(os.environ.get("MSB_DECK", "N12M"))
decks = np.stack([deck, deck])
W0 = np.random.RandomState(2024).uniform(0, 1, 10)
t0 = time.time()
eng = BatchEngine(n, lanes_per_game=U)
seeds = np.arange(n, dtype=np.uint32)
eng.reset(seeds, decks)
print("reset ok", time.time() - t0, flush=True)
orc = oracle_lib.Oracle(n)
for i in range(n):
    assert orc.reset(i, i, deck, deck) == 0
bad = sum(eng.export(i) != orc.canon(i) for i in range(n))
print("initial state mismatches:", bad, flush=True)
assert bad == 0
# random policy via the step API
pol = np.random.RandomState(1)
for t in range(40):
    masks = eng.legal_mask()
    acts = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        assert (masks[i] == orc.legal_mask(i)).all(), (t, i)
        la = eng.legal_actions(mask=masks[i])
        acts[i] = la[pol.randint(0, len(la))]
    r, d, f = eng.step(acts)
    for i in range(n):
        fo, ro, do = orc.step(i, acts[i])
        assert (fo, ro, do) == (f[i], r[i], d[i]), (t, i, acts[i], (fo, ro, do), (f[i], r[i], d[i]))
    bad = sum(eng.export(i) != orc.canon(i) for i in range(n))
    assert bad == 0, (t, bad)
print("40 random-policy rounds bit-exact", time.time() - t0, flush=True)
obs, raises = eng.observe()
feat = eng.features()
for i in range(n):
    assert np.array_equal(obs[i], orc.observe(i))
    assert np.array_equal(feat[i].view(np.uint64), orc.features(i).view(np.uint64))
print("observation + features bit-exact", flush=True)
# heuristic decisions
for t in range(30):
    a, best, scores = eng.decide(W0, want_scores=True)
    for i in range(n):
        if orc.have_winner(i):
            assert a[i] == 255
            continue
        ao, so, _ = orc.decide(i, W0)
        assert a[i] == ao, (t, i, a[i], ao)
        assert np.array_equal(scores[i].view(np.uint64), so.view(np.uint64)), (t, i)
        orc.step(i, ao)
    bad = sum(eng.export(i) != orc.canon(i) for i in range(n))
    assert bad == 0, (t, bad)
print("30 heuristic decision rounds bit-exact (scores, argmax, committed state)", time.time() - t0, flush=True)
print(eng.stats())
