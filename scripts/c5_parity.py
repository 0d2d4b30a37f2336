#!/usr/bin/env python3
"""Configuration C5 at scale against the CPU replay (test infrastructure: uses oracle/ through tests/oracle_rollout.py).

    gpurun -- python scripts/c5_parity.py [individuals=4096] [games_per_individual=128]

One generation through FitnessEvaluator.evaluate_population: per-game decks drawn on the device from the 109 observable
cards (monsoon_draw_decks), every game on the smallest record its decks need, overflowing games replayed on the next larger
one; then the same schedule on the 16-thread CPU replay (decks by the CPU restatement of the draw).  Prints the timings, the
tiers and whether every fitness value, result, decision count and fault code agrees; exit status 1 on any difference.
tests/test_gpu_parity.py::test_config_c5_one_generation_full_size is this at the default size."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle_rollout import oracle_draw_decks, oracle_rollout_fn_mt  # noqa: E402
from monsoon_amd.cards import RANDOM_DECK  # noqa: E402
from monsoon_amd.config import EvolutionaryConfig  # noqa: E402
from monsoon_amd.fitness import FitnessEvaluator, record_limited  # noqa: E402
from monsoon_amd.weights import WeightVector  # noqa: E402

n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gpi = int(sys.argv[2]) if len(sys.argv) > 2 else 128
np.random.seed(11)
pop = [WeightVector(10) for _ in range(n_ind)]
cfg = EvolutionaryConfig(mu=n_ind, lambda_=n_ind, schedule="ring", games_per_individual=gpi, deck=RANDOM_DECK, max_turns=200, max_concurrent_games=65536)
hip = FitnessEvaluator(cfg)
hip.use_hall_of_fame = False
hip.evaluate_population(pop[:min(64, n_ind)], generation=0)   # creates the engines of both tiers
hip.reset_stats()
hip.tier_games, hip.capacity_replays, hip.capacity_faults, hip.depth_faults = [0, 0], 0, 0, 0
t0 = time.time()
f_hip = hip.evaluate_population(pop, generation=3)
t_hip = time.time() - t0
results, steps, faults = hip.last_rollout
kms, launches = hip.kernel_time()
box = {}


def cpu_rollout(weights, matches, deck_pairs, max_turns):
    box["rows"] = oracle_rollout_fn_mt(weights, matches, deck_pairs, max_turns, want_faults=True)
    return box["rows"][0]


cpu = FitnessEvaluator(cfg, rollout_fn=cpu_rollout, deck_draw_fn=oracle_draw_decks)
cpu.use_hall_of_fame = False
t0 = time.time()
f_cpu = cpu.evaluate_population(pop, generation=3)
t_cpu = time.time() - t0
_, ores, osteps, of = box["rows"]
bad = int((np.array(f_hip) != np.array(f_cpu)).sum() + (results != ores).sum() + (steps != osteps).sum() + (faults != of).sum())
st = hip.get_stats()
codes, cnt = np.unique(faults, return_counts=True)
print(f"C5 generation, {n_ind} x {gpi} = {n_ind * gpi} games: GPU {t_hip:.2f} s end to end (k_play {kms / 1e3:.2f} s in {launches} launches, "
      f"{st['env_steps'] / 1e6:.0f} M env-steps = {st['env_steps'] / t_hip / 1e6:.0f} M/s), CPU replay on 16 threads {t_cpu:.1f} s")
print(f"  tiers: {hip.tier_games[0]} games on the standard record, {hip.tier_games[1]} on the extended one, {hip.capacity_replays} replayed on a larger "
      f"record, {int(record_limited(faults).sum())} left on a record limit, {int((faults == 18).sum())} ended by the recursion guard")
print(f"  fault codes {dict(zip(codes.tolist(), cnt.tolist()))}; rows that differ from the CPU replay: {bad}")
sys.exit(1 if bad else 0)
