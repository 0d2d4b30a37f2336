#!/usr/bin/env python3
"""Config C3: the GA loop with N = 256 evaluated individuals per generation (mu = lambda = 128;
generation 0 evaluates mu), 64 games per individual as FIRST player against (i+1+k) mod N, deck N12M both
sides, max_turns 200, config.seed 42, 10 generations, one MI355X.  Prints one JSON line."""
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from monsoon_amd.config import EvolutionaryConfig  # noqa: E402
from monsoon_amd.evolution import EvolutionEngine  # noqa: E402

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = EvolutionaryConfig(mu=128, lambda_=128, generations=gens, schedule="ring", games_per_individual=64, deck="N12M",
                         max_turns=200, seed=42, checkpoint_interval=1000, save_logs=True,
                         results_dir=tempfile.mkdtemp(prefix="c3_"))
eng = EvolutionEngine(cfg)
eng.initialize()
t0 = time.time()
res = eng.run()
dt = time.time() - t0
st = res["evaluation_stats"]
print(json.dumps({"config": "C3", "generations": res["generations"], "wall_s": dt, "games": st["total_games"],
                  "env_steps": st["env_steps"], "env_steps_per_s": st["env_steps"] / st["total_time"],
                  "games_per_s": st["games_per_second"], "best_fitness": res["best_fitness"],
                  "mean_fitness": res["final_stats"]["mean_fitness"],
                  "evaluate_ms_per_generation": [round(1e3 * t, 1) for t in eng.fitness_evaluator.eval_times]}))
