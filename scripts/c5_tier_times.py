#!/usr/bin/env python3
"""Kernel time of one C5 generation by record tier (GPU box): python scripts/c5_tier_times.py [individuals=4096] [games=128]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from monsoon_amd.cards import RANDOM_DECK  # noqa: E402
from monsoon_amd.config import EvolutionaryConfig  # noqa: E402
from monsoon_amd.fitness import FitnessEvaluator  # noqa: E402
from monsoon_amd.weights import WeightVector  # noqa: E402

n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gpi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
np.random.seed(11)
pop = [WeightVector(10) for _ in range(n_ind)]
cfg = EvolutionaryConfig(mu=n_ind, lambda_=n_ind, schedule="ring", games_per_individual=gpi, deck=RANDOM_DECK, max_turns=200, max_concurrent_games=65536)
ev = FitnessEvaluator(cfg)
ev.use_hall_of_fame = False
ev.evaluate_population(pop[:min(64, n_ind)], generation=0)
ev.reset_stats()
t0 = time.time()
ev.evaluate_population(pop, generation=3)
dt = time.time() - t0
names = {0: "standard", 1: "extended", 2: "large"}
print(f"C5 generation {n_ind} x {gpi}: {dt:.2f} s end to end; tiers first played {ev.tier_games}, replayed {ev.capacity_replays}")
for t, e in sorted(ev._engines.items()):
    ms, n = e.kernel_time()
    print(f"  {names[t]:9s} record: k_play {ms:8.1f} ms in {n} launches, {e.stats()['lookahead_steps'] / 1e6:7.1f} M look-ahead steps")
