"""World-size-2 gloo test of the N>1 path: the schedule is sharded by row individual, each rank
plays its shard (oracle backend on CPU), and one all_reduce sums the per-individual counters.
The sharded result must equal the single-process one."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from monsoon_amd.config import EvolutionaryConfig
from monsoon_amd.fitness import FitnessEvaluator
from monsoon_amd.weights import WeightVector


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _population():
    np.random.seed(11)
    return [WeightVector(10) for _ in range(4)]


def _deck_schedule():
    from monsoon_amd.cards import DECKS
    from monsoon_amd.decks import DeckEvolutionConfig
    # generation 3 is in the explore phase: a different deck pair per game, drawn from one seeded stream
    return DeckEvolutionConfig(DECKS["IRONCLAD"], DECKS["SWARM"], exploit_generations=1, explore_generations=4, seed=9)


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_rollout import oracle_rollout_fn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = EvolutionaryConfig(mu=4, lambda_=4, schedule="ring", games_per_individual=2, max_turns=10)
    ev = FitnessEvaluator(cfg, rollout_fn=oracle_rollout_fn)
    f = ev.evaluate_population(_population(), generation=3)
    np.save(os.path.join(out_dir, f"fit{rank}.npy"), np.array(f))
    # with a deck schedule every rank draws the decks of the WHOLE schedule and keeps its shard's
    ev2 = FitnessEvaluator(cfg, _deck_schedule(), rollout_fn=oracle_rollout_fn)
    f2 = ev2.evaluate_population(_population(), generation=3)
    np.save(os.path.join(out_dir, f"fitdeck{rank}.npy"), np.array(f2))
    dist.destroy_process_group()


def test_sharded_evaluation_matches_single_process(tmp_path):
    from oracle_rollout import oracle_rollout_fn
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0 = np.load(tmp_path / "fit0.npy")
    f1 = np.load(tmp_path / "fit1.npy")
    cfg = EvolutionaryConfig(mu=4, lambda_=4, schedule="ring", games_per_individual=2, max_turns=10)
    single = FitnessEvaluator(cfg, rollout_fn=oracle_rollout_fn).evaluate_population(_population(), generation=3)
    assert np.array_equal(f0, f1)
    assert np.array_equal(f0, np.array(single))
    d0 = np.load(tmp_path / "fitdeck0.npy")
    d1 = np.load(tmp_path / "fitdeck1.npy")
    single_deck = FitnessEvaluator(cfg, _deck_schedule(), rollout_fn=oracle_rollout_fn).evaluate_population(_population(), generation=3)
    assert np.array_equal(d0, d1)
    assert np.array_equal(d0, np.array(single_deck))
