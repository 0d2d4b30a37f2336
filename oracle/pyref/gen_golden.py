#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the Python reference itself.

TEST INFRASTRUCTURE (build container only; needs /root/reference, numpy 2.2.6).
The reference has no fixtures of its own for this path, so its behaviour is pinned
by running it here and committing inputs + expected outputs (data only, no reference source):

  rng_kat.npz              G1  numpy RandomState known answers (u32, random, randint, shuffle)
  score_kat.npz            G4  HeuristicAgent score for random (w, f_before, f_after) via np.dot
  trace_random_<deck>.npz  G3  seeded random-policy games: legal masks, action, state hash,
                               observation hash, reward, done (+ features for N12M)
  trace_pool.npz           G3/G5 random 12-card decks from the 107 cards of the standard record (card coverage)
  trace_pool_ext.npz       G3/G5 the same over all 109 observable cards (ua20, b005: extended record)
  trace_pool_up.npz        G6 decks holding up01 / up02 / up03, whose int(card) makes get_observation raise
  trace_expert.npz         (f1) both sides driven by Stormbound.expert_action (the bot's choices are the actions)
  trace_heuristic_N12M.npz G3  corrected heuristic self-play (SURVEY §8c contract), W0 both sides:
                               chosen action, best score, score hash, state hash per decision
  initial_states.npz       G2  canonical records right after construction
  population_seed42.npz        GA driver: initial population + first offspring for config.seed=42

Usage: PYTHONHASHSEED=0 python oracle/pyref/gen_golden.py [--only NAME] [--jobs N]
"""
import argparse
import time
import contextlib
import io
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness as H  # noqa: E402

GOLD = os.path.join(H.REPO, "tests", "golden")
UNSUPPORTED = {"ua20", "b005"}
FAULT_CARDS = {"up01", "up02", "up03"}
W0 = np.random.RandomState(2024).uniform(0, 1, 10)


def idx(deck):
    return np.array([H.CARD_INDEX[c] for c in deck], dtype=np.uint8)


def obs_hash(obs):
    return H.fnv1a64(np.ascontiguousarray(obs, dtype="<i4").tobytes())


# ---------------------------------------------------------------------------------------------
def gen_rng():
    seeds = [0, 1, 42, 123, 2**32 - 1]
    out = {"seeds": np.array(seeds, dtype=np.uint64)}
    bounds = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 20, 22, 100, 1000] * 8, dtype=np.int32)
    out["randint_bounds"] = bounds
    for s in seeds:
        rs = np.random.RandomState(s)
        out[f"u32_{s}"] = rs.randint(0, 4294967296, size=1500, dtype=np.uint32)
        rs = np.random.RandomState(s)
        out[f"random_{s}"] = np.array([rs.random() for _ in range(400)])
        rs = np.random.RandomState(s)
        out[f"randint_{s}"] = np.array([rs.randint(0, int(b)) for b in bounds], dtype=np.int32)
        rs = np.random.RandomState(s)
        sh = []
        for _ in range(20):
            a = list(range(12))
            rs.shuffle(a)
            sh.append(a)
        out[f"shuffle12_{s}"] = np.array(sh, dtype=np.int32)
    np.savez_compressed(os.path.join(GOLD, "rng_kat.npz"), **out)


def gen_score():
    from evo.heuristic_agent import HeuristicAgent
    from evo.weights import WeightVector

    class F:
        def __init__(self, v):
            self.v = v
            self.mana_efficiency = v[0]

        def get_feature_vector(self):
            return self.v

    rs = np.random.RandomState(7)
    n = 2000
    w = rs.uniform(0, 1, (n, 10))
    fb = np.zeros((n, 10))
    fa = np.zeros((n, 10))
    scores = np.zeros(n)
    for i in range(n):
        # feature-like magnitudes: ratios in [0,1], integer-ish strengths, weighted sums k/5
        fb[i] = [rs.uniform(0, 1), rs.randint(-20, 21), rs.uniform(-1, 1), rs.randint(-4, 5) / 4.0, rs.randint(-60, 60),
                 rs.randint(-8, 9), rs.randint(-3, 4), rs.randint(0, 400) / 5.0, rs.randint(0, 400) / 5.0, rs.uniform(0, 1)]
        fa[i] = [rs.uniform(0, 1), rs.randint(-20, 21), rs.uniform(-1, 1), rs.randint(-4, 5) / 4.0, rs.randint(-60, 60),
                 rs.randint(-8, 9), rs.randint(-3, 4), rs.randint(0, 400) / 5.0, rs.randint(0, 400) / 5.0, rs.uniform(0, 1)]
        wv = WeightVector(10)
        wv.weights = w[i].copy()
        ag = HeuristicAgent(wv, 0)
        b, a = F(fb[i]), F(fa[i])
        agent = ag._compute_feature_delta(b, a, for_agent=True)
        enemy = ag._compute_feature_delta(b, a, for_agent=False)
        res = ag._compute_resource_delta(b, a)
        scores[i] = enemy - agent - res
    np.savez_compressed(os.path.join(GOLD, "score_kat.npz"), w=w, before=fb, after=fa, score=scores)


# ---------------------------------------------------------------------------------------------
def random_trace(args):
    seed, d0, d1, steps, want_feat = args[:5]
    expert = len(args) > 5 and args[5]
    from evo.features import StateFeatures
    g = H.make_game(seed, d0, d1)
    pol = np.random.RandomState(seed + 1000)
    rec = dict(legal=[], action=[], hash=[], obs=[], reward=[], done=[], feat=[], fault=0)
    init_hash = H.fnv1a64(H.canon(g))
    for _ in range(steps):
        la = g.legal_actions()
        try:
            # expert traces: both sides are Stormbound.expert_action (games/stormbound.py:563-637)
            a = int(g.expert_action()) if expert else int(la[pol.randint(0, len(la))])
        except Exception:  # noqa: BLE001  random.choice([]) inside the bot: the trace ends here
            rec["legal"].append(H.legal_mask(la))
            rec["action"].append(255)
            rec["fault"] = 1
            break
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                obs, reward, done = g.step(a)
        except Exception:  # noqa: BLE001  the agent layer swallows these; the trace ends here
            rec["legal"].append(H.legal_mask(la))
            rec["action"].append(a)
            rec["fault"] = 1
            break
        rec["legal"].append(H.legal_mask(la))
        rec["action"].append(a)
        rec["hash"].append(H.fnv1a64(H.canon(g)))
        rec["obs"].append(obs_hash(obs))
        rec["reward"].append(reward)
        rec["done"].append(int(done))
        if want_feat:
            rec["feat"].append(StateFeatures(obs, g.to_play()).get_feature_vector())
        if g.have_winner():
            break
    return seed, init_hash, rec


def pack_traces(name, jobs, results, decks0, decks1):
    out = dict(seeds=[], init_hash=[], offsets=[0], fault=[], legal=[], action=[], hash=[], obs=[], reward=[], done=[], feat=[])
    for seed, init_hash, rec in results:
        out["seeds"].append(seed)
        out["init_hash"].append(init_hash)
        out["fault"].append(rec["fault"])
        out["legal"] += rec["legal"]
        out["action"] += rec["action"]
        # faulted final step has no post-state: pad so that every array is indexed by step
        pad = len(rec["action"]) - len(rec["hash"])
        out["hash"] += rec["hash"] + [0] * pad
        out["obs"] += rec["obs"] + [0] * pad
        out["reward"] += rec["reward"] + [0] * pad
        out["done"] += rec["done"] + [0] * pad
        out["feat"] += rec["feat"]
        out["offsets"].append(len(out["action"]))
    arrs = dict(
        seeds=np.array(out["seeds"], dtype=np.uint32), init_hash=np.array(out["init_hash"], dtype=np.uint64),
        offsets=np.array(out["offsets"], dtype=np.int64), fault=np.array(out["fault"], dtype=np.uint8),
        legal=np.array(out["legal"], dtype=np.uint64).reshape(-1, 3), action=np.array(out["action"], dtype=np.uint8),
        hash=np.array(out["hash"], dtype=np.uint64), obs=np.array(out["obs"], dtype=np.uint64),
        reward=np.array(out["reward"], dtype=np.int8), done=np.array(out["done"], dtype=np.uint8),
        deck0=np.array(decks0, dtype=np.uint8), deck1=np.array(decks1, dtype=np.uint8))
    if out["feat"]:
        arrs["feat"] = np.array(out["feat"], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, name), **arrs)
    print(name, "games", len(out["seeds"]), "steps", len(out["action"]))


def gen_random(deck, deck2, n_games, steps, jobs, want_feat=False, seed0=0):
    d0, d1 = H.DECKS[deck], H.DECKS[deck2 or deck]
    tasks = [(seed0 + k, d0, d1, steps, want_feat) for k in range(n_games)]
    with ProcessPoolExecutor(jobs) as ex:
        results = list(ex.map(random_trace, tasks))
    pack_traces(f"trace_random_{deck}.npz", jobs, results, [idx(d0)] * n_games, [idx(d1)] * n_games)


def gen_expert(n_games, steps, jobs):
    pool = [c for c in H.CARD_IDS if c not in UNSUPPORTED and c not in FAULT_CARDS]
    tasks, decks0, decks1 = [], [], []
    for k in range(n_games):
        seed = 60000 + k
        if k < n_games // 2:
            d0, d1 = H.DECKS["IRONCLAD"], H.DECKS["SWARM"]
        else:
            rs = np.random.RandomState(seed ^ 0x9E3779B9)
            d0 = list(rs.choice(pool, 12, replace=False))
            d1 = list(rs.choice(pool, 12, replace=False))
        tasks.append((seed, d0, d1, steps, False, True))
        decks0.append(idx(d0))
        decks1.append(idx(d1))
    with ProcessPoolExecutor(jobs) as ex:
        results = list(ex.map(random_trace, tasks))
    pack_traces("trace_expert.npz", jobs, results, decks0, decks1)


def gen_pool(n_games, steps, jobs, ext=False):
    pool = [c for c in H.CARD_IDS if (ext or c not in UNSUPPORTED) and c not in FAULT_CARDS]
    tasks, decks0, decks1 = [], [], []
    for k in range(n_games):
        seed = (30000 if ext else 5000) + k
        rs = np.random.RandomState(seed ^ 0x9E3779B9)
        d0 = list(rs.choice(pool, 12, replace=False))
        d1 = list(rs.choice(pool, 12, replace=False))
        tasks.append((seed, d0, d1, steps, False))
        decks0.append(idx(d0))
        decks1.append(idx(d1))
    with ProcessPoolExecutor(jobs) as ex:
        results = list(ex.map(random_trace, tasks))
    pack_traces("trace_pool_ext.npz" if ext else "trace_pool.npz", jobs, results, decks0, decks1)


def gen_pool_up(n_games, steps, jobs):
    """G6: decks holding up01 / up02 / up03, whose int(card) raises (card.py:46): get_observation -- and with it
    Stormbound.step -- raises as soon as one of them is visible to the mover (own hand / deck, board, last-4 history).
    A third of the games hold one in the FIRST player's deck (the very first step raises), the rest only in the second
    player's (the first PASS, or the card appearing in the history / on the board, raises)."""
    pool = [c for c in H.CARD_IDS if c not in UNSUPPORTED and c not in FAULT_CARDS]
    ups = sorted(FAULT_CARDS)
    tasks, decks0, decks1 = [], [], []
    for k in range(n_games):
        seed = 42000 + k
        rs = np.random.RandomState(seed ^ 0x9E3779B9)
        d0 = list(rs.choice(pool, 12, replace=False))
        d1 = list(rs.choice(pool, 12, replace=False))
        (d0 if k % 3 == 0 else d1)[int(rs.randint(0, 12))] = ups[k % 3]
        if k % 5 == 4:
            d1[0], d1[1], d1[2] = ups   # all three at once
        tasks.append((seed, d0, d1, steps, False))
        decks0.append(idx(d0))
        decks1.append(idx(d1))
    with ProcessPoolExecutor(jobs) as ex:
        results = list(ex.map(random_trace, tasks))
    pack_traces("trace_pool_up.npz", jobs, results, decks0, decks1)


# ---------------------------------------------------------------------------------------------
def heuristic_trace(args):
    """Corrected rollout loop: while not have_winner() and steps < max_turns."""
    seed, d0, d1, max_turns = args[:4]
    w_second = args[4] if len(args) > 4 else None   # a different weight vector for the SECOND player's agent
    from evo.game_adapter import StormboundAdapter
    from evo.heuristic_agent import HeuristicAgent
    from evo.weights import WeightVector
    from games.stormbound import Game

    game = Game.__new__(Game)
    game.env = H.make_game(seed, d0, d1)
    wv = WeightVector(10)
    wv.weights = W0.copy()
    wv1 = wv
    if w_second is not None:
        wv1 = WeightVector(10)
        wv1.weights = np.array(w_second, dtype=np.float64)
    agents = [HeuristicAgent(wv, 0), HeuristicAgent(wv1, 1)]
    adapter = StormboundAdapter(game)
    rec = dict(action=[], hash=[], best=[], shash=[], nlegal=[])
    steps = 0
    fault = 0
    with contextlib.redirect_stdout(io.StringIO()):
        while not adapter.game.env.have_winner() and steps < max_turns:
            agent = agents[adapter.get_current_player()]
            legal = adapter.get_legal_actions()
            scores = np.array([agent.score_action(adapter, a) for a in legal], dtype=np.float64)
            k = int(np.argmax(scores))
            a = legal[k]
            rec["action"].append(a)
            rec["best"].append(scores[k])
            rec["shash"].append(H.fnv1a64(scores.tobytes()))
            rec["nlegal"].append(len(legal))
            try:
                adapter = adapter.apply_action(a)
            except Exception:   # evo/fitness.py:208-210: the exception ends the game as a draw
                rec["hash"].append(0)
                fault = 1
                steps += 1
                break
            steps += 1
            rec["hash"].append(H.fnv1a64(H.canon(adapter.game.env)))
    env = adapter.game.env
    b = {int(env.board.local.order): env.board.local.strength, int(env.board.remote.order): env.board.remote.strength}
    result = -1 if fault else (0 if (b[1] < 0 <= b[0]) else 1 if (b[0] < 0 <= b[1]) else -1)
    return seed, rec, result, fault


def gen_heuristic(n_games, max_turns, jobs, deck0="N12M", deck1=None, seed0=0, two_weights=False):
    d0, d1 = H.DECKS[deck0], H.DECKS[deck1 or deck0]
    w1 = np.random.RandomState(7).uniform(0, 1, 10) if two_weights else None
    tasks = [(seed0 + s, d0, d1, max_turns) + ((w1,) if two_weights else ()) for s in range(n_games)]
    with ProcessPoolExecutor(jobs) as ex:
        results = list(ex.map(heuristic_trace, tasks))
    out = dict(seeds=[], offsets=[0], result=[], fault=[], action=[], hash=[], best=[], shash=[], nlegal=[])
    for seed, rec, result, fault in results:
        out["seeds"].append(seed)
        out["result"].append(result)
        out["fault"].append(fault)
        for k in ("action", "hash", "best", "shash", "nlegal"):
            out[k] += rec[k]
        out["offsets"].append(len(out["action"]))
    name = f"trace_heuristic_{deck0}{'_2w' if two_weights else ''}.npz"
    np.savez_compressed(
        os.path.join(GOLD, name), seeds=np.array(out["seeds"], dtype=np.uint32),
        offsets=np.array(out["offsets"], dtype=np.int64), result=np.array(out["result"], dtype=np.int8),
        fault=np.array(out["fault"], dtype=np.uint8),
        action=np.array(out["action"], dtype=np.uint8), hash=np.array(out["hash"], dtype=np.uint64),
        best=np.array(out["best"], dtype=np.float64), shash=np.array(out["shash"], dtype=np.uint64),
        nlegal=np.array(out["nlegal"], dtype=np.int16), deck=idx(d0), deck1=idx(d1), w0=W0,
        w1=(w1 if two_weights else W0), max_turns=np.int32(max_turns))
    print(name, "games", n_games, "decisions", len(out["action"]), "look-ahead steps", int(np.sum(out["nlegal"])),
          "ended by an exception", int(np.sum(out["fault"])))


def gen_heuristic_pool(n_games, max_turns, jobs, seed0=700, ext=False):
    """Heuristic self-play on per-game random 12-card decks (standard-build pool): look-aheads and committed steps
    that raise in the reference are common here (evo/heuristic_agent.py:48-51 -> score 0.0, evo/fitness.py:208-210)."""
    pool = [c for c in H.CARD_IDS if c not in (("up01", "up02", "up03") if ext else ("ua20", "b005", "up01", "up02", "up03"))]
    tasks, decks = [], []
    for s in range(n_games):
        rs = np.random.RandomState((seed0 + s) ^ 0x9E3779B9)
        d0 = [str(c) for c in rs.choice(pool, 12, replace=False)]
        d1 = [str(c) for c in rs.choice(pool, 12, replace=False)]
        tasks.append((seed0 + s, d0, d1, max_turns))
        decks.append([idx(d0), idx(d1)])
    with ProcessPoolExecutor(jobs) as ex:
        results = list(ex.map(heuristic_trace, tasks))
    out = dict(seeds=[], offsets=[0], result=[], fault=[], action=[], hash=[], best=[], shash=[], nlegal=[])
    for seed, rec, result, fault in results:
        out["seeds"].append(seed)
        out["result"].append(result)
        out["fault"].append(fault)
        for k in ("action", "hash", "best", "shash", "nlegal"):
            out[k] += rec[k]
        out["offsets"].append(len(out["action"]))
    np.savez_compressed(
        os.path.join(GOLD, "trace_heuristic_pool_ext.npz" if ext else "trace_heuristic_pool.npz"), seeds=np.array(out["seeds"], dtype=np.uint32),
        offsets=np.array(out["offsets"], dtype=np.int64), result=np.array(out["result"], dtype=np.int8),
        fault=np.array(out["fault"], dtype=np.uint8),
        action=np.array(out["action"], dtype=np.uint8), hash=np.array(out["hash"], dtype=np.uint64),
        best=np.array(out["best"], dtype=np.float64), shash=np.array(out["shash"], dtype=np.uint64),
        nlegal=np.array(out["nlegal"], dtype=np.int16), decks=np.array(decks, dtype=np.uint8), w0=W0, max_turns=np.int32(max_turns))
    print("trace_heuristic_pool_ext.npz" if ext else "trace_heuristic_pool.npz", "games", n_games, "decisions", len(out["action"]), "look-ahead steps", int(np.sum(out["nlegal"])),
          "ended by an exception", int(np.sum(out["fault"])))


def gen_heuristic_c5(indices, max_turns, jobs):
    """Heuristic self-play of chosen games of the C5 family (tests/c5_games.py: seed 90000 + k, decks drawn by
    RandomState(k ^ 0x9E3779B9) from the 109 observable cards): games whose nested b005 memories outgrow the product's
    extended record -- the product replays them on its large record (libmonsoon_hip_big.so) and must reproduce these.
    29409 is a game the reference itself ends with a RecursionError (the product: FAULT_DEPTH at the same decision); 1374
    takes the reference more than ten minutes and is left out by the time limit."""
    pool = [c for c in H.CARD_IDS if c not in ("up01", "up02", "up03")]
    tasks, decks = [], []
    for k in indices:
        rs = np.random.RandomState(k ^ 0x9E3779B9)
        d0 = [str(c) for c in rs.choice(pool, 12, replace=False)]
        d1 = [str(c) for c in rs.choice(pool, 12, replace=False)]
        tasks.append((90000 + k, d0, d1, max_turns))
        decks.append([idx(d0), idx(d1)])
    # one process per game with a time limit: the reference's nested deep copies make a few of these games take hours
    import multiprocessing as mp
    limit = float(os.environ.get("C5_GAME_SECONDS", "600"))
    ctx = mp.get_context("fork")
    procs = []
    for k, t in zip(indices, tasks):
        q = ctx.Queue()
        p = ctx.Process(target=lambda q=q, t=t: q.put(heuristic_trace(t)))
        p.start()
        procs.append((k, p, q))
    results, kept, kept_decks = [], [], []
    t_end = time.time() + limit
    for (k, p, q), d in zip(procs, decks):
        try:
            r = q.get(timeout=max(1.0, t_end - time.time()))
            results.append(r)
            kept.append(k)
            kept_decks.append(d)
        except Exception:   # queue.Empty
            print("game", k, "not finished by the reference within", limit, "s: left out")
        p.join(timeout=1.0)
        if p.is_alive():
            p.terminate()
    indices, decks = kept, kept_decks
    out = dict(seeds=[], offsets=[0], result=[], fault=[], action=[], hash=[], best=[], shash=[], nlegal=[])
    for seed, rec, result, fault in results:
        out["seeds"].append(seed)
        out["result"].append(result)
        out["fault"].append(fault)
        for k in ("action", "hash", "best", "shash", "nlegal"):
            out[k] += rec[k]
        out["offsets"].append(len(out["action"]))
    np.savez_compressed(
        os.path.join(GOLD, os.environ.get("C5_OUT", "trace_heuristic_c5_big.npz")), seeds=np.array(out["seeds"], dtype=np.uint32),
        offsets=np.array(out["offsets"], dtype=np.int64), result=np.array(out["result"], dtype=np.int8),
        fault=np.array(out["fault"], dtype=np.uint8),
        action=np.array(out["action"], dtype=np.uint8), hash=np.array(out["hash"], dtype=np.uint64),
        best=np.array(out["best"], dtype=np.float64), shash=np.array(out["shash"], dtype=np.uint64),
        nlegal=np.array(out["nlegal"], dtype=np.int16), decks=np.array(decks, dtype=np.uint8), w0=W0, max_turns=np.int32(max_turns),
        games=np.array(indices, dtype=np.int32))
    print("trace_heuristic_c5_big.npz games", len(indices), "decisions", len(out["action"]), "look-ahead steps", int(np.sum(out["nlegal"])),
          "ended by an exception", int(np.sum(out["fault"])))


def gen_initial():
    recs, lens, seeds = [], [], []
    for deck in ("N12V", "N12M", "S12"):
        for seed in (0, 1, 42, 123, 2**32 - 1):
            c = H.canon(H.make_game(seed, H.DECKS[deck], H.DECKS[deck]))
            buf = np.zeros(2048, dtype=np.uint8)
            buf[:len(c)] = np.frombuffer(c, dtype=np.uint8)
            recs.append(buf)
            lens.append(len(c))
            seeds.append(seed)
    np.savez_compressed(os.path.join(GOLD, "initial_states.npz"), canon=np.array(recs), length=np.array(lens, dtype=np.int32),
                        seeds=np.array(seeds, dtype=np.uint64),
                        decks=np.array([idx(H.DECKS[d]) for d in ("N12V", "N12M", "S12") for _ in range(5)], dtype=np.uint8))


def gen_population():
    """GA driver numerics: Population.initialize_population + generate_offspring with config.seed=42."""
    from evo.config import EvolutionaryConfig
    from evo.population import Population
    cfg = EvolutionaryConfig(mu=12, lambda_=12, seed=42)
    with contextlib.redirect_stdout(io.StringIO()):
        pop = Population(cfg)
        pop.initialize_population(10)
        w0 = np.array([i.get_weights() for i in pop.individuals])
        s0 = np.array([i.get_sigmas() for i in pop.individuals])
        off = pop.generate_offspring()
    np.savez_compressed(os.path.join(GOLD, "population_seed42.npz"), init_weights=w0, init_sigmas=s0,
                        off_weights=np.array([i.get_weights() for i in off]), off_sigmas=np.array([i.get_sigmas() for i in off]))


def gen_decks():
    """utils.generate_random_deck / DeckEvolutionConfig of the reference, Python's global `random` seeded per case."""
    import json
    import random
    import utils as RU
    from enums import Faction
    cls = {c: getattr(H.refcards, c.upper()) for c in H.CARD_IDS}

    def ids(deck):
        return [type(c).__name__.lower() for c in deck]

    cases = []
    arche = {"IRONCLAD": H.DECKS["IRONCLAD"], "SWARM": H.DECKS["SWARM"], "N12M": H.DECKS["N12M"]}
    k = 0
    for fac in (0, 1, 2, 3, 4):
        for name, deck in arche.items():
            for ratio in (0.0, 0.25, 0.5, 0.9, 1.0):
                k += 1
                random.seed(1000 + k)
                out = RU.generate_random_deck(Faction(fac), original=[cls[c]() for c in deck], preserve_ratio=ratio)
                cases.append({"seed": 1000 + k, "faction": fac, "original": list(deck), "ratio": ratio, "deck": ids(out)})
    sched = []
    for seed in (7, 8):
        cfg = RU.DeckEvolutionConfig([cls[c]() for c in H.DECKS["IRONCLAD"]], [cls[c]() for c in H.DECKS["SWARM"]],
                                     exploit_generations=2, explore_generations=4, max_random_ratio=0.5, balance_archetype_ratio=0.7)
        random.seed(seed)
        rows = []
        for gen in range(10):
            for _ in range(3):
                d1, d2 = cfg.get_deck_configuration(gen)
                rows.append({"generation": gen, "p1": ids(d1), "p2": ids(d2)})
        sched.append({"seed": seed, "rows": rows, "phase": [cfg.get_phase_info(g) for g in range(10)]})
    with open(os.path.join(GOLD, "deck_schedule.json"), "w") as f:
        json.dump({"generate_random_deck": cases, "schedule": sched}, f, indent=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--jobs", type=int, default=6)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    todo = {
        "rng": gen_rng,
        "score": gen_score,
        "initial": gen_initial,
        "population": gen_population,
        "random_N12V": lambda: gen_random("N12V", None, 16, 300, args.jobs),
        "random_N12M": lambda: gen_random("N12M", None, 48, 300, args.jobs, want_feat=True),
        "random_IRONCLAD": lambda: gen_random("IRONCLAD", "SWARM", 32, 300, args.jobs),
        "pool": lambda: gen_pool(160, 300, args.jobs),
        "pool_ext": lambda: gen_pool(120, 300, args.jobs, ext=True),   # all 109 observable cards (ua20, b005 included)
        "pool_up": lambda: gen_pool_up(30, 300, args.jobs),   # up01/up02/up03: int(card) raises (G6)
        "random_S12": lambda: gen_random("S12", None, 48, 300, args.jobs),
        "expert": lambda: gen_expert(48, 300, args.jobs),
        "heuristic": lambda: gen_heuristic(12, 200, args.jobs),
        "heuristic_2w": lambda: gen_heuristic(8, 150, args.jobs, "N12M", None, 1200, two_weights=True),
        "heuristic_S12": lambda: gen_heuristic(16, 200, args.jobs, "S12", None, 300),
        "heuristic_pool": lambda: gen_heuristic_pool(24, 120, args.jobs),
        "heuristic_pool_ext": lambda: gen_heuristic_pool(16, 120, args.jobs, 900, ext=True),
        "heuristic_c5_big": lambda: gen_heuristic_c5([int(k) for k in os.environ["C5_GAMES"].split(",")] if os.environ.get("C5_GAMES") else
                                                     [264, 1374, 2103, 2458, 2649, 3525, 3540, 6149, 3691, 4215, 4593, 5070, 29409], 200, args.jobs),
        "heuristic_IRONCLAD": lambda: gen_heuristic(12, 120, args.jobs, "IRONCLAD", "SWARM", 400),
        "decks": gen_decks,
    }
    for name, fn in todo.items():
        if args.only and args.only != name:
            continue
        fn()
        print("done", name)


if __name__ == "__main__":
    main()
