"""CPU tests of the host side that stays Python: config, GA operators, schedules, Seam F."""
import json
import os

import numpy as np
import pytest

from monsoon_amd.config import EvolutionaryConfig
from monsoon_amd.evolution import EvolutionEngine
from monsoon_amd.fitness import (FitnessEvaluator, fitness_from_counts, hash32, ring_schedule, round_robin_schedule,
                                 shard_by_individual)
from monsoon_amd.population import Population
from monsoon_amd.weights import WeightVector
from oracle_rollout import oracle_rollout_fn


def test_config_validation_and_json(tmp_path):
    with pytest.raises(ValueError):
        EvolutionaryConfig(mu=0)
    with pytest.raises(ValueError):
        EvolutionaryConfig(tau=-1)
    nested = {"population": {"mu": 4, "lambda_": 6}, "evolution": {"generations": 3},
              "evaluation": {"games_per_pairing": 2, "deck_configs": 1},
              "simulation": {"max_turns": 50, "num_workers": 2, "timeout_seconds": 5}}
    p = tmp_path / "c.json"
    p.write_text(json.dumps(nested))
    c = EvolutionaryConfig.from_json(str(p))
    assert (c.mu, c.lambda_, c.generations, c.games_per_pairing, c.max_turns) == (4, 6, 3, 2, 50)
    assert EvolutionaryConfig.from_dict(c.to_dict()).to_dict() == c.to_dict()


def test_population_matches_reference_numerics(gold):
    """Same numpy global-stream call order as evo/population.py: seed 42 gives the reference's
    initial population and first offspring (fixture generated from the reference)."""
    g = gold("population_seed42.npz")
    pop = Population(EvolutionaryConfig(mu=12, lambda_=12, seed=42))
    pop.initialize_population(10)
    assert np.array_equal(np.array([i.get_weights() for i in pop.individuals]), g["init_weights"])
    assert np.array_equal(np.array([i.get_sigmas() for i in pop.individuals]), g["init_sigmas"])
    off = pop.generate_offspring()
    assert np.array_equal(np.array([i.get_weights() for i in off]), g["off_weights"])
    assert np.array_equal(np.array([i.get_sigmas() for i in off]), g["off_sigmas"])


def test_weight_vector_bounds():
    np.random.seed(0)
    w = WeightVector(10)
    for _ in range(50):
        w.mutate(0.5, 0.5, 1e-5)
        assert (w.weights >= 0).all() and (w.weights <= 1).all() and (w.sigmas >= 1e-5).all()
    c = w.copy()
    assert np.array_equal(c.weights, w.weights) and c.weights is not w.weights


def test_schedules():
    m = round_robin_schedule(3, 5, 2, generation=7)
    assert len(m) == 3 * 4 * 2
    assert not np.any((m["p1"] == m["p2"]))
    assert set(m["p2"]) == {0, 1, 2, 3, 4} and set(m["p1"]) == {0, 1, 2}
    assert len(set(m["seed"])) == len(m)
    r = ring_schedule(8, 3, generation=0)
    assert len(r) == 24 and all(r["p2"][i] == (r["p1"][i] + 1 + i % 3) % 8 for i in range(24))
    parts = [shard_by_individual(r, 8, k, 3) for k in range(3)]
    assert sum(len(p) for p in parts) == len(r)
    assert hash32(1, 2, 3) == hash32(1, 2, 3) != hash32(3, 2, 1)
    assert fitness_from_counts(np.array([[3, 2, 8]]), 8) == [0.5]


def test_as_written_mode_reproduces_reference_logs():
    """The reference's fitness loop never plays a game: every individual scores exactly 1.0
    (results/evolutionary2/training_log.csv: best = mean = 1.0, std = 0)."""
    np.random.seed(1)
    cfg = EvolutionaryConfig(mu=4, lambda_=4, games_per_pairing=3, mode="as_written")
    ev = FitnessEvaluator(cfg)
    pop = [WeightVector(10) for _ in range(4)]
    assert ev.evaluate_population(pop, 0) == [1.0] * 4
    assert len(ev.hall_of_fame) == 4
    assert ev.evaluate_population(pop, 1) == [1.0] * 4      # now with hall-of-fame opponents
    assert ev.get_stats()["total_games"] == 4 * 3 * 3 + 4 * 7 * 3


def test_evaluator_rollout_mode_with_oracle_backend():
    np.random.seed(2)
    cfg = EvolutionaryConfig(mu=3, lambda_=3, games_per_pairing=1, max_turns=12)
    ev = FitnessEvaluator(cfg, rollout_fn=oracle_rollout_fn)
    pop = [WeightVector(10) for _ in range(3)]
    f = ev.evaluate_population(pop, 0)
    assert len(f) == 3 and all(0.0 <= x <= 1.0 for x in f)
    # 12 decisions cannot finish an N12M game: all draws -> 0.5
    assert f == [0.5, 0.5, 0.5]


def test_evolution_engine_end_to_end(tmp_path):
    cfg = EvolutionaryConfig(mu=3, lambda_=3, generations=2, games_per_pairing=1, max_turns=6, seed=5,
                             checkpoint_interval=1, results_dir=str(tmp_path / "res"))
    eng = EvolutionEngine(cfg, rollout_fn=oracle_rollout_fn)
    eng.initialize()
    res = eng.run()
    assert res["generations"] == 2
    log = open(os.path.join(cfg.results_dir, "training_log.csv")).read().splitlines()
    assert log[0].startswith("generation,time,best_fitness") and len(log) == 3
    assert os.path.exists(res["final_population_file"])
    eng2 = EvolutionEngine(cfg, rollout_fn=oracle_rollout_fn)
    eng2.load_checkpoint(res["final_population_file"])
    assert eng2.population.generation == 2


def test_vectorised_schedules_equal_the_scalar_definition():
    """hash32_array / the numpy schedules give exactly the rows of the per-match Python loops they replace."""
    from monsoon_amd.fitness import MATCH_DTYPE, hash32, hash32_array, ring_schedule, round_robin_schedule
    assert [int(x) for x in hash32_array(7, np.arange(5), 3)] == [hash32(7, i, 3) for i in range(5)]
    ref = np.array([(i, (i + 1 + k) % 16, hash32(3, i, k), 0) for i in range(16) for k in range(8)], dtype=MATCH_DTYPE)
    assert (ring_schedule(16, 8, 3) == ref).all()
    rows = [(i, j, hash32(5, i, j, g), 0) for i in range(6) for j in range(9) if not (j < 6 and i == j) for g in range(2)]
    assert (round_robin_schedule(6, 9, 2, 5) == np.array(rows, dtype=MATCH_DTYPE)).all()


def test_deck_schedule_equals_the_reference_draws():
    """monsoon_amd.decks vs utils.generate_random_deck / DeckEvolutionConfig of the reference run with the same
    `random` seed (tests/golden/deck_schedule.json, oracle/pyref/gen_golden.py --only decks)."""
    import json
    import random
    from monsoon_amd.cards import DECKS
    from monsoon_amd.decks import DeckEvolutionConfig, generate_random_deck
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "deck_schedule.json")))
    assert len(g["generate_random_deck"]) == 75
    for c in g["generate_random_deck"]:
        assert generate_random_deck(c["faction"], c["original"], c["ratio"], random.Random(c["seed"])) == c["deck"]
    for s in g["schedule"]:
        cfg = DeckEvolutionConfig(DECKS["IRONCLAD"], DECKS["SWARM"], 2, 4, 0.5, 0.7, seed=s["seed"])
        rows = iter(s["rows"])
        for gen in range(10):
            for _ in range(3):
                r = next(rows)
                assert cfg.get_deck_configuration(gen) == (r["p1"], r["p2"])
        assert [cfg.get_phase_info(k) for k in range(10)] == s["phase"]


def test_evaluator_draws_a_deck_pair_per_game_from_the_schedule():
    """Seam F with a deck schedule: one pair for the whole generation while exploiting, a fresh draw per game
    afterwards (games/evolutionary_stormbound.py:52); decks reach the rollout as index arrays."""
    from monsoon_amd.cards import DECKS, deck_indices
    from monsoon_amd.decks import DeckEvolutionConfig
    seen = []

    def fake(weights, matches, deck_pairs, max_turns):
        seen.append((np.array(matches["deck"]), np.array(deck_pairs)))
        c = np.zeros((len(weights), 3), dtype=np.int64)
        np.add.at(c[:, 2], matches["p1"], 1)
        return c

    np.random.seed(3)
    cfg = EvolutionaryConfig(mu=3, lambda_=3, games_per_pairing=2, max_turns=5)
    pop = [WeightVector(10) for _ in range(3)]
    for trial in range(2):
        dc = DeckEvolutionConfig(DECKS["IRONCLAD"], DECKS["SWARM"], exploit_generations=1, explore_generations=2, seed=11)
        ev = FitnessEvaluator(cfg, dc, rollout_fn=fake)
        ev.use_hall_of_fame = False
        ev.evaluate_population(pop, 0)
        ev.evaluate_population(pop, 2)
    (d0, p0), (d2, p2) = seen[0], seen[1]
    assert p0.shape == (1, 2, 12) and (d0 == 0).all()
    assert np.array_equal(p0[0, 0], deck_indices("IRONCLAD")) and np.array_equal(p0[0, 1], deck_indices("SWARM"))
    assert p2.shape == (12, 2, 12) and np.array_equal(d2, np.arange(12))
    assert len({p2[k].tobytes() for k in range(12)}) > 1            # explore phase: the games differ
    assert np.array_equal(seen[3][1], p2)                          # same seed, same schedule


def test_every_card_with_an_ability_has_a_case():
    """ability_cases.inc (one function per card) against the card table generated from the reference's constructors:
    every unit/structure that overrides activate_ability and every spell has exactly one entry, nothing else has."""
    import json
    import re
    root = os.path.join(os.path.dirname(__file__), "..", "monsoon_amd")
    cases = open(os.path.join(root, "csrc", "ability_cases.inc")).read()
    # MSB_CARD / MSB_SPELL: leaf abilities; the _K forms: abilities that make a nested call (they take the work stack)
    entity = [m.lower()[2:] for m in re.findall(r"MSB_CARD(?:_K)?\((C_\w+),", cases)]
    spells = [m.lower()[2:] for m in re.findall(r"MSB_SPELL(?:_K)?\((C_\w+),", cases)]
    meta = json.load(open(os.path.join(root, "card_ids.json")))
    assert sorted(entity) == sorted(c["id"] for c in meta if c["kind"] != 2 and c["has_ability"])
    assert sorted(spells) == sorted(c["id"] for c in meta if c["kind"] == 2)
    assert len(set(entity)) == len(entity) and len(set(spells)) == len(spells)


def test_random109_decks_and_per_game_tiers_on_the_cpu_stand_in():
    """Configuration C5's deck rule through the evaluator (CPU stand-in for the rollout): per-game decks from the game's own
    pre-stream -- the CPU restatement of the draw gives the same schedule as numpy itself --, every game on the smallest
    record its decks need, and the tiered rows equal the rows of the same schedule played on the extended record alone
    wherever no record limit was met."""
    from oracle_rollout import oracle_draw_decks, oracle_rollout_fn, oracle_rollout_tier
    from monsoon_amd.cards import RANDOM_DECK, needs_extended_each
    from monsoon_amd.config import EvolutionaryConfig
    from monsoon_amd.fitness import FitnessEvaluator
    from monsoon_amd.weights import WeightVector
    np.random.seed(3)
    pop = [WeightVector(10) for _ in range(6)]
    cfg = EvolutionaryConfig(mu=6, lambda_=6, schedule="ring", games_per_individual=5, deck=RANDOM_DECK, max_turns=40)
    seen = []

    def spy(weights, matches, deck_pairs, max_turns):
        seen.append((np.array(matches), np.array(deck_pairs)))
        return oracle_rollout_fn(weights, matches, deck_pairs, max_turns)
    f_numpy = FitnessEvaluator(cfg, rollout_fn=spy).evaluate_population(pop, 2)
    f_orc = FitnessEvaluator(cfg, rollout_fn=spy, deck_draw_fn=oracle_draw_decks).evaluate_population(pop, 2)
    assert f_numpy == f_orc and np.array_equal(seen[0][1], seen[1][1]) and seen[0][1].shape == (30, 2, 12)
    m, pairs = seen[0]
    assert np.array_equal(m["deck"], np.arange(30)) and all(len(set(d.tolist())) == 12 for d in pairs.reshape(-1, 12))
    ext = needs_extended_each(pairs)
    assert ext.any() and not ext.all()
    w = np.stack([p.weights for p in pop])
    _, r_t, s_t, f_t = oracle_rollout_fn(w, m, pairs, 40, want_faults=True)
    _, r_e, s_e, f_e = oracle_rollout_tier(w, m, pairs, 40, 1)
    ok = (f_t < 16) & (f_e < 16)
    assert ok.all() and np.array_equal(r_t, r_e) and np.array_equal(s_t, s_e) and np.array_equal(f_t, f_e)
    # the two sub-schedules from two host threads (what the GPU path does with its two handles): the same rows
    from monsoon_amd.fitness import tiered_rollout
    c_c, r_c, s_c, f_c, _, sizes = tiered_rollout(lambda tier, sub, sp: oracle_rollout_tier(w, sub, sp, 40, tier), len(w), m, pairs, concurrent=True)
    assert sizes == [int((~ext).sum()), int(ext.sum())] and np.array_equal(r_c, r_t) and np.array_equal(s_c, s_t) and np.array_equal(f_c, f_t)


def test_recursion_guard_is_not_a_record_limit_and_strict_mode_raises():
    """fitness.tiered_rollout: a game ended by the recursion guard (code 18: the reference's RecursionError, the same on every
    record) is never replayed; a game left on a record limit after the largest tier is reported -- a warning, or an error
    with strict=True."""
    import warnings
    from monsoon_amd.config import EvolutionaryConfig
    from monsoon_amd.fitness import MATCH_DTYPE, FitnessEvaluator, record_limited, tiered_rollout
    assert record_limited(np.array([0, 1, 16, 18, 22], dtype=np.uint8)).tolist() == [False, False, True, False, True]
    m = np.zeros(4, dtype=MATCH_DTYPE)
    m["seed"] = np.arange(4)
    pairs = np.zeros((1, 2, 12), dtype=np.uint8)
    calls = []

    def play(tier, sub, sub_pairs):
        calls.append((tier, sub["seed"].tolist()))
        f = np.array([{0: 18, 1: 16, 2: 0, 3: 23}[int(s)] if tier < 2 else (16 if s == 3 else 0) for s in sub["seed"]], dtype=np.uint8)
        r = np.where(f != 0, -1, 0).astype(np.int8)
        c = np.zeros((1, 3), dtype=np.int64)
        c[0] = [(r == 0).sum(), (r == -1).sum(), len(sub)]
        return c, r, np.full(len(sub), 7, dtype=np.int32), f
    counts, results, steps, faults, replays, sizes = tiered_rollout(play, 1, m, pairs)
    assert calls == [(0, [0, 1, 2, 3]), (1, [1, 3]), (2, [1, 3])] and replays == 4 and sizes == [4, 0]
    assert faults.tolist() == [18, 0, 0, 16] and counts[0].tolist() == [2, 2, 4]
    cfg = EvolutionaryConfig(max_turns=5)
    for strict in (False, True):
        fe = FitnessEvaluator(cfg, strict=strict)
        fe._engine = lambda tier: None
        import monsoon_amd.fitness as F
        orig = F.tiered_rollout
        F.tiered_rollout = lambda play_, n, mm, pp, concurrent=False: tiered_rollout(play, n, mm, pp, concurrent=concurrent)
        try:
            if strict:
                import pytest
                with pytest.raises(RuntimeError, match="still end on a limit"):
                    fe._hip_rollout(np.zeros((1, 10)), m, pairs, 5)
            else:
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    fe._hip_rollout(np.zeros((1, 10)), m, pairs, 5)
                assert fe.capacity_faults == 1 and fe.depth_faults == 1 and any("still end on a limit" in str(x.message) for x in w)
        finally:
            F.tiered_rollout = orig


def test_single_process_evaluation_does_not_import_torch_distributed():
    """A process that is not a rank of a torch.distributed job never imports it (it cost generation 0 of a C3 run 0.7 s:
    profiles/r03_run_c3.json); the sharded path is covered by tests/test_distributed_cpu.py."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{os.path.dirname(here)!r}, {here!r}]\n"
        "from monsoon_amd.config import EvolutionaryConfig\n"
        "from monsoon_amd.fitness import FitnessEvaluator\n"
        "from monsoon_amd.weights import WeightVector\n"
        "np.random.seed(1)\n"
        "pop = [WeightVector(10) for _ in range(4)]\n"
        "cfg = EvolutionaryConfig(mu=4, lambda_=4, schedule='ring', games_per_individual=2, deck='N12M', max_turns=5)\n"
        "ev = FitnessEvaluator(cfg, rollout_fn=lambda w, m, d, t: np.tile(np.array([0, 0, 1]), (len(w), 1)))\n"
        "f = ev.evaluate_population(pop, generation=0)\n"
        "print(len(f), 'torch.distributed' in sys.modules, 'torch' in sys.modules)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["4", "False", "False"], out.stdout
