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
    # configuration C5's per-game random decks depend on the game's seed alone: every rank draws only the decks of its shard
    from oracle_rollout import oracle_draw_decks
    drawn = []

    def draw(seeds, pool):
        drawn.append(len(seeds))
        return oracle_draw_decks(seeds, pool)
    cfg3 = EvolutionaryConfig(mu=4, lambda_=4, schedule="ring", games_per_individual=2, max_turns=10, deck="random109")
    f3 = FitnessEvaluator(cfg3, rollout_fn=oracle_rollout_fn, deck_draw_fn=draw).evaluate_population(_population(), generation=3)
    np.save(os.path.join(out_dir, f"fitrand{rank}.npy"), np.array(f3))
    np.save(os.path.join(out_dir, f"drawn{rank}.npy"), np.array(drawn))
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
    from oracle_rollout import oracle_draw_decks
    cfg3 = EvolutionaryConfig(mu=4, lambda_=4, schedule="ring", games_per_individual=2, max_turns=10, deck="random109")
    single_rand = FitnessEvaluator(cfg3, rollout_fn=oracle_rollout_fn, deck_draw_fn=oracle_draw_decks).evaluate_population(_population(), generation=3)
    r0, r1 = np.load(tmp_path / "fitrand0.npy"), np.load(tmp_path / "fitrand1.npy")
    assert np.array_equal(r0, r1) and np.array_equal(r0, np.array(single_rand))
    assert np.load(tmp_path / "drawn0.npy").tolist() == [4] and np.load(tmp_path / "drawn1.npy").tolist() == [4]   # 8 games, 4 per rank


def _run_bench(args, env_extra):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no rendezvous in the environment starts two ranks (torch.distributed.run, one
    process per device) and reports the rank count the all-reduce saw.  CPU ranks: gloo backend and bench.py's
    stand-in engine (MONSOON_BENCH_FAKE, a test hook) -- what is exercised is the launcher and the rank plumbing."""
    env = {"MONSOON_BENCH_FAKE": "1", "MONSOON_BENCH_BACKEND": "gloo"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        assert k not in os.environ
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--games", "64", "--no-cpu"], env)
    assert rc == 0, err[-2000:]
    assert len(lines) == 1   # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks"] == {"launched": 2, "in_all_reduce": 2, "backend": "gloo"}
    assert line["steps"] == 3 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["lookahead_per_decision"] - 17.0) < 1e-9   # both ranks' counters were summed


def test_bench_refuses_more_gpus_than_the_machine_has():
    """On a machine with fewer GPUs than --gpus (this container has none) the bench fails loudly instead of printing a
    line with a smaller n_gpus."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"], {})
    assert rc != 0 and lines == []
    assert "refusing" in err


def test_replace_capacity_faulted_rewrites_only_the_overflowing_rows():
    """Host logic of the replay tier (monsoon_amd/fitness.py): rows with a fault code >= 16 are replaced by the replay's,
    their first attempt's contribution to the per-individual counts is taken out, everything else is untouched."""
    import numpy as np
    from monsoon_amd.fitness import MATCH_DTYPE, replace_capacity_faulted
    m = np.zeros(6, dtype=MATCH_DTYPE)
    m["p1"] = [0, 0, 1, 1, 2, 2]
    m["p2"] = [1, 2, 0, 2, 0, 1]
    results = np.array([0, -1, 1, -1, 0, -1], dtype=np.int8)      # games 1 and 3 were cut short as draws ...
    steps = np.array([50, 7, 60, 9, 70, 200], dtype=np.int32)
    faults = np.array([0, 16, 0, 22, 1, 0], dtype=np.uint8)       # ... by capacity codes; code 1 is the reference's own exception
    counts = np.zeros((3, 3), dtype=np.int64)
    for k in range(6):
        counts[m["p1"][k], 0] += results[k] == 0
        counts[m["p1"][k], 1] += results[k] == -1
        counts[m["p1"][k], 2] += 1
    seen = {}

    def replay(sub):
        seen["sub"] = sub.copy()
        c = np.zeros((3, 3), dtype=np.int64)
        r = np.array([0, 1], dtype=np.int8)                        # replayed: game 1 is a win for p1 = 0, game 3 a loss for p1 = 1
        for j in range(2):
            c[sub["p1"][j], 0] += r[j] == 0
            c[sub["p1"][j], 1] += r[j] == -1
            c[sub["p1"][j], 2] += 1
        return c, r, np.array([120, 130], dtype=np.int32), np.array([0, 16], dtype=np.uint8)

    n = replace_capacity_faulted(counts, results, steps, faults, m, replay)
    assert n == 2 and list(seen["sub"]["p2"]) == [2, 2]
    assert results.tolist() == [0, 0, 1, 1, 0, -1] and steps.tolist() == [50, 120, 60, 130, 70, 200]
    assert faults.tolist() == [0, 0, 0, 16, 1, 0]                  # what not even the replay could hold stays flagged
    assert counts.tolist() == [[2, 0, 2], [0, 0, 2], [1, 1, 2]]
    assert replace_capacity_faulted(counts, results, steps, np.zeros(6, dtype=np.uint8), m, replay) == 0
