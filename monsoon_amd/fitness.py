"""FitnessEvaluator -- Seam F: evaluate_population(population, generation) -> List[float]
(mirror of evo/fitness.py:18-259, the only call the GA driver makes into the hot path,
evo/evolution.py:88).

The reference's rollout loop is dead code (StormboundAdapter.is_terminal is always True,
evo/game_adapter.py:339-342 -> every game is scored as a player-1 win in 0 steps); it is kept as
mode="as_written".  mode="rollout" plays the games for real under the contract of SURVEY.md §8c:
    while not have_winner() and steps < max_turns: a = agent[to_play].select_action(); apply(a)
    P1 wins iff P2's base < 0 <= P1's base; P2 wins iff P1's base < 0 <= P2's base; else draw;
    a faulting committed step ends the game as a draw (evo/fitness.py:170-174, 208-210).
All games of a generation form one schedule of (p1, p2, seed, deck) entries which the HIP engine
plays monsoon_config.max_games at a time.  With torch.distributed initialised the schedule is
sharded by row individual, one process per GPU, and the per-individual {wins, draws, games}
counters are summed with one all_reduce (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import time

import numpy as np

MATCH_DTYPE = np.dtype([("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])


def hash32(*vals):
    """Deterministic game seed from (generation, i, j, game) -- the reference uses Game(seed=None)
    (evo/fitness.py:144-145), i.e. OS entropy; a reproducible build needs an explicit seed."""
    h = 0x9E3779B9
    for v in vals:
        h ^= (int(v) + 0x7F4A7C15 + ((h << 6) & 0xFFFFFFFF) + (h >> 2)) & 0xFFFFFFFF
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        h ^= h >> 16
    return h


def hash32_array(*vals):
    """hash32 over broadcastable integer arrays (same values as the scalar form, element by element)."""
    M = np.uint64(0xFFFFFFFF)
    vals = np.broadcast_arrays(*[np.asarray(v, dtype=np.uint64) for v in vals])
    h = np.full(vals[0].shape, 0x9E3779B9, dtype=np.uint64)
    for v in vals:
        h = h ^ ((v + np.uint64(0x7F4A7C15) + ((h << np.uint64(6)) & M) + (h >> np.uint64(2))) & M)
        h = (h * np.uint64(0x85EBCA6B)) & M
        h = h ^ (h >> np.uint64(13))
        h = (h * np.uint64(0xC2B2AE35)) & M
        h = h ^ (h >> np.uint64(16))
    return h.astype(np.uint32)


def round_robin_schedule(n_individuals, n_total_opponents, games_per_pairing, generation):
    """evo/fitness.py:53-59,133: every individual vs every other opponent (population + hall of fame)."""
    i, j, g = np.meshgrid(np.arange(n_individuals), np.arange(n_total_opponents), np.arange(games_per_pairing), indexing="ij")
    keep = (i != j).ravel()   # j < n_individuals and i == j is skipped; a hall-of-fame j is never < n_individuals when equal
    i, j, g = i.ravel()[keep], j.ravel()[keep], g.ravel()[keep]
    out = np.zeros(len(i), dtype=MATCH_DTYPE)
    out["p1"], out["p2"], out["seed"] = i, j, hash32_array(generation, i, j, g)
    return out


def ring_schedule(n_individuals, games_per_individual, generation):
    """individual i plays FIRST against (i+1+k) mod N, k = 0..games-1."""
    i, k = np.meshgrid(np.arange(n_individuals), np.arange(games_per_individual), indexing="ij")
    i, k = i.ravel(), k.ravel()
    out = np.zeros(len(i), dtype=MATCH_DTYPE)
    out["p1"], out["p2"], out["seed"] = i, (i + 1 + k) % n_individuals, hash32_array(generation, i, k)
    return out


def shard_by_individual(matches, n_individuals, rank, world):
    """Contiguous blocks of row individuals per rank."""
    lo = (n_individuals * rank) // world
    hi = (n_individuals * (rank + 1)) // world
    return matches[(matches["p1"] >= lo) & (matches["p1"] < hi)]


def fitness_from_counts(counts, games_per_individual):
    """wins + 0.5*draws, normalised (evo/fitness.py:111-113,160-166)."""
    counts = np.asarray(counts, dtype=np.float64)
    return [float((counts[i, 0] + 0.5 * counts[i, 1]) / games_per_individual) for i in range(len(counts))]


CAPACITY_CODE = 16   # fault codes >= this are limits of a build's record (csrc/msb_base.h), not reference behaviour ...
DEPTH_CODE = 18      # ... except the recursion guard: 40 nested abilities / moves.  Where that trips the reference's own
#                      recursion ends in RecursionError (raising the guard to 200 changes no game; the reference's trace of such
#                      a game is part of tests/golden/trace_heuristic_c5_big.npz), and the guard is the same on every record: such
#                      a game is a draw like any other exception, it is not replayed and not counted as a record limit


def record_limited(faults):
    """bool[n]: the games a LARGER record could play further (capacity codes other than the recursion guard)."""
    faults = np.asarray(faults)
    return (faults >= CAPACITY_CODE) & (faults != DEPTH_CODE)


def replace_capacity_faulted(counts, results, steps, faults, matches, replay):
    """The games of a rollout that hit a limit of the record (fault code >= 16: the reference's deep copies nest without
    bound, a record does not) are played again by `replay(sub_matches) -> (counts, results, steps, faults)` -- the same
    games on the build with the larger record -- and the rows of the first attempt replaced.  Returns the number of
    games replayed; the arrays are updated in place."""
    bad = np.nonzero(faults >= CAPACITY_CODE)[0]
    if len(bad) == 0:
        return 0
    c2, r2, s2, f2 = replay(matches[bad])
    p1 = matches["p1"][bad]
    np.subtract.at(counts[:, 0], p1, results[bad] == 0)
    np.subtract.at(counts[:, 1], p1, results[bad] == -1)
    np.subtract.at(counts[:, 2], p1, 1)
    counts += np.asarray(c2, dtype=counts.dtype)
    results[bad], steps[bad], faults[bad] = r2, s2, f2
    return len(bad)


def tiered_rollout(play, n_rows, matches, deck_pairs, concurrent=False):
    """A schedule played on the smallest record each game needs.  play(tier, sub_matches, sub_deck_pairs) ->
    (counts[n_rows][3], results, steps, faults) runs matches on one build (tier 0 standard, 1 extended, 2 large record);
    it is handed only the deck pairs its matches name (a build refuses a table holding a card it does not support).

    Tier per GAME, not per call: only a deck pair holding ua20 / b005 needs the extended record (2 400 bytes against
    752: the hot kernel runs half as many wavefronts with half as many candidate lanes on it), so a schedule of random
    109-card decks -- 37 % of whose games hold one of the two -- is split in two sub-schedules.  Then the ladder: games
    that hit a limit of their record (fault code >= 16) are played again on the next larger one and their rows replaced
    (nested b005 memories are deep copies of the whole game, cards/b005.py:14-33, card.py:71-75: the reference's copies
    nest without bound, a record does not).  concurrent: the standard and the extended sub-schedule are played from two
    host threads (two handles, a stream each: the wavefronts of one fill the GPU while the other's batch drains or its
    host side works).  Returns (counts, results, steps, faults, replays, tier_sizes)."""
    from .cards import needs_extended_each
    matches = np.asarray(matches)
    deck_pairs = np.asarray(deck_pairs, dtype=np.uint8).reshape(-1, 2, 12)
    ext_pair = needs_extended_each(deck_pairs)

    def run(t, sub):
        used, inv = np.unique(sub["deck"], return_inverse=True)
        sub = sub.copy()
        sub["deck"] = inv
        return play(t, sub, deck_pairs[used])
    tier = ext_pair[matches["deck"]].astype(np.int8) if len(ext_pair) > 1 else np.full(len(matches), int(ext_pair[0]), dtype=np.int8)
    counts = np.zeros((n_rows, 3), dtype=np.int64)
    results = np.zeros(len(matches), dtype=np.int8)
    steps = np.zeros(len(matches), dtype=np.int32)
    faults = np.zeros(len(matches), dtype=np.uint8)
    first = [np.nonzero(tier == t)[0] for t in (0, 1)]
    sizes = [len(idx) for idx in first]
    todo = [(t, idx) for t, idx in enumerate(first) if len(idx)]
    if concurrent and len(todo) == 2:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=2) as pool:
            done = list(pool.map(lambda ti: run(ti[0], matches[ti[1]]), todo))
    else:
        done = [run(t, matches[idx]) for t, idx in todo]
    for (t, idx), (c, r, s_, f) in zip(todo, done):
        counts += np.asarray(c, dtype=np.int64)
        results[idx], steps[idx], faults[idx] = r, s_, f
    replays = 0
    for t in (1, 2):   # standard -> extended -> large
        bad = np.nonzero(record_limited(faults) & (tier < t))[0]
        if not len(bad):
            continue
        replays += len(bad)
        c2, r2, s2, f2 = run(t, matches[bad])
        p1 = matches["p1"][bad]
        np.subtract.at(counts[:, 0], p1, results[bad] == 0)
        np.subtract.at(counts[:, 1], p1, results[bad] == -1)
        np.subtract.at(counts[:, 2], p1, 1)
        counts += np.asarray(c2, dtype=np.int64)
        results[bad], steps[bad], faults[bad] = r2, s2, f2
        tier[bad] = t
    return counts, results, steps, faults, replays, sizes


class FitnessEvaluator:
    def __init__(self, config, deck_config=None, rollout_fn=None, device=None, deck_draw_fn=None, strict=False):
        self.config = config
        self.deck_config = deck_config      # monsoon_amd.decks.DeckEvolutionConfig (utils.py:121-242) or None = config.deck both sides
        self.total_games = 0
        self.total_time = 0.0
        self.total_env_steps = 0
        self.total_decisions = 0
        self.hall_of_fame = []
        self.hall_of_fame_size = 5
        self.use_hall_of_fame = True
        self._rollout_fn = rollout_fn
        self._deck_draw_fn = deck_draw_fn   # test hook like rollout_fn: (pre-stream seeds, pool) -> uint8[n][2][12]
        self._device = device
        self._engines = {}
        self.eval_times = []            # wall seconds of every evaluate_population call (the first one creates the engine)
        self.strict = strict            # True: a game left on a record limit raises instead of being scored as a draw
        self.depth_faults = 0           # games ended by the recursion guard (the reference's RecursionError), draws
        self.capacity_replays = 0       # games replayed on the large record
        self.capacity_faults = 0        # games not even the large record could hold (their fault code ends them as draws)
        self.tier_games = [0, 0]        # games first played on the standard / the extended record

    # -- device ------------------------------------------------------------------------------
    def _engine(self, tier):
        from .engine import BatchEngine
        if self._engines.get(tier) is None:
            dev = self._device
            if dev is None:
                import os
                dev = int(os.environ.get("LOCAL_RANK", "0"))
            # the large record is a replay tier for a few games per thousand: a small handle, its default variant
            games = self.config.max_concurrent_games if tier < 2 else min(self.config.max_concurrent_games, 2048)
            self._engines[tier] = BatchEngine(games, device=dev, lanes_per_game=self.config.lanes_per_game if tier == 0 else 0, extended=tier)
        return self._engines[tier]

    def warm_up(self):
        """Create the standard-record engine and launch once now instead of inside the first evaluate_population call; the
        extended / large engines are still created when a schedule first needs them."""
        if self._rollout_fn is None and self.config.mode == "rollout":
            eng = self._engine(0)
            # one decision of as many games as fill the GPU: loads the code objects and makes the runtime size the
            # launch-time buffers (work-stack overflow blocks, the queue's scratch) for a full grid -- 0.8 s the first time
            from .cards import deck_indices
            d = deck_indices("N12M")
            n = min(self.config.max_concurrent_games, 8192)
            m = np.zeros(n, dtype=MATCH_DTYPE)
            m["seed"] = np.arange(n)
            eng.rollout(np.zeros((1, 10)), m, np.stack([d, d])[None], 1)
            eng.reset_stats()

    def _hip_rollout(self, weights, matches, deck_pairs, max_turns):
        import threading
        lock = threading.Lock()

        def play(tier, sub, sub_pairs):
            eng = self._engine(tier)
            before = eng.stats()["lookahead_steps"]
            counts, results, steps = eng.rollout(weights, sub, sub_pairs, max_turns, want_results=True)
            with lock:
                self.total_env_steps += eng.stats()["lookahead_steps"] - before
                self.total_decisions += int(steps.sum())
            return counts.astype(np.int64), results, steps, eng.rollout_faults(len(sub))

        counts, results, steps, faults, replays, sizes = tiered_rollout(play, len(weights), matches, deck_pairs,
                                                                        concurrent=self.config.concurrent_tiers)
        self.capacity_replays += replays
        left = int(record_limited(faults).sum())
        self.capacity_faults += left
        self.depth_faults += int((faults == DEPTH_CODE).sum())
        if left:
            msg = (f"{left} of {len(matches)} games still end on a limit of the largest record (fault codes "
                   f"{sorted(set(faults[record_limited(faults)].tolist()))}): scored as draws, parity with the reference unpinned for them")
            if self.strict:
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        self.tier_games = [a + b for a, b in zip(self.tier_games, sizes)]
        self.last_rollout = (results, steps, faults)
        return counts

    def _decks_for(self, matches, generation):
        """Deck pairs [n_decks][2][12] for a schedule; fills matches["deck"].  Without a deck_config: config.deck both
        sides.  With one: utils.py:155-219 per GAME, as games/evolutionary_stormbound.py:52 draws them (one pair for
        the whole generation while the schedule is in its exploit phase)."""
        from .cards import C5_STREAM_XOR, RANDOM_DECK, deck_indices, draw_random_decks_numpy, observable_pool
        if self.deck_config is None and self.config.deck == RANDOM_DECK:
            # configuration C5: two decks per game from the 109 observable cards, drawn by the game's own pre-stream
            pre = matches["seed"] ^ np.uint32(C5_STREAM_XOR)
            matches["deck"] = np.arange(len(matches))
            if self._deck_draw_fn is not None:
                return self._deck_draw_fn(pre, observable_pool())
            if self._rollout_fn is not None:   # CPU stand-in (tests): numpy itself
                return draw_random_decks_numpy(pre)
            return self._engine(0).draw_decks(pre, observable_pool())
        if self.deck_config is None:
            deck = deck_indices(self.config.deck)
            return np.stack([deck, deck])[None]
        if self.deck_config.is_static(generation):
            d1, d2 = self.deck_config.get_deck_configuration(generation)
            return np.stack([deck_indices(d1), deck_indices(d2)])[None]
        pairs = np.zeros((len(matches), 2, 12), dtype=np.uint8)
        for k in range(len(matches)):
            d1, d2 = self.deck_config.get_deck_configuration(generation)
            pairs[k, 0], pairs[k, 1] = deck_indices(d1), deck_indices(d2)
        matches["deck"] = np.arange(len(matches))
        return pairs

    @staticmethod
    def _dist():
        """torch.distributed if this process is a rank of an initialised job of more than one rank, else None.  A process
        that never imported torch.distributed cannot be one: it is not imported here (0.7 s the first time)."""
        import sys
        dist = sys.modules.get("torch.distributed")
        if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
        return None

    # -- Seam F ------------------------------------------------------------------------------
    def evaluate_population(self, population, generation=0):
        from .cards import deck_indices
        cfg = self.config
        n = len(population)
        opponents = list(population)
        if self.use_hall_of_fame and self.hall_of_fame:
            opponents.extend(self.hall_of_fame)
        n_total = len(opponents)
        start = time.time()
        if cfg.schedule == "ring":
            matches = ring_schedule(n, cfg.games_per_individual, generation)
            per_individual = cfg.games_per_individual
        else:
            matches = round_robin_schedule(n, n_total, cfg.games_per_pairing, generation)
            per_individual = (n_total - 1) * cfg.games_per_pairing
        if cfg.mode == "as_written":
            # _play_game returns 0 ("agent1 wins") without a step for every game
            counts = np.zeros((n_total, 3), dtype=np.int64)
            np.add.at(counts[:, 0], matches["p1"], 1)
            np.add.at(counts[:, 2], matches["p1"], 1)
        else:
            weights = np.stack([np.asarray(o.weights, dtype=np.float64) for o in opponents])
            from .cards import RANDOM_DECK
            dist = self._dist()
            mine = matches
            if self.deck_config is None and cfg.deck == RANDOM_DECK:
                # per-game decks that depend on the game's seed alone: every rank draws only the games it plays
                if dist is not None:
                    mine = shard_by_individual(matches, n, dist.get_rank(), dist.get_world_size()).copy()
                deck_pairs = self._decks_for(mine, generation) if len(mine) else np.zeros((1, 2, 12), dtype=np.uint8)
            else:
                deck_pairs = self._decks_for(matches, generation)   # a sequential stream (utils.py:155-219): drawn for the WHOLE schedule, so every rank sees the same decks
                if dist is not None:
                    mine = shard_by_individual(matches, n, dist.get_rank(), dist.get_world_size())
                    if len(deck_pairs) > 1 and len(mine):   # keep only this rank's decks
                        deck_pairs = deck_pairs[mine["deck"]]
                        mine = mine.copy()
                        mine["deck"] = np.arange(len(mine))
            fn = self._rollout_fn or self._hip_rollout
            counts = np.zeros((n_total, 3), dtype=np.int64)
            if len(mine):
                counts += np.asarray(fn(weights, mine, deck_pairs, cfg.max_turns), dtype=np.int64)
            if dist is not None:
                counts = self._all_reduce_counts(dist, counts)
        self.total_games += len(matches)
        self.total_time += time.time() - start
        self.eval_times.append(time.time() - start)
        fitness = fitness_from_counts(counts[:n], per_individual)
        self._update_hall_of_fame(population, fitness)
        return fitness

    @staticmethod
    def _all_reduce_counts(dist, counts):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(counts))
        if dist.get_backend() == "nccl":   # RCCL: the tensor must live on this rank's GPU
            t = t.cuda()
        dist.all_reduce(t)   # sum over ranks; <= 48 KB at N = 4096
        return t.cpu().numpy()

    # evo/fitness.py:247-259: copies of the five best of this generation
    def _update_hall_of_fame(self, population, fitness):
        ranked = sorted(zip(fitness, population), key=lambda p: p[0], reverse=True)
        self.hall_of_fame = [ind.copy() for _, ind in ranked[:self.hall_of_fame_size]]

    def get_stats(self):
        return {"total_games": self.total_games, "total_time": self.total_time,
                "avg_time_per_game": self.total_time / max(self.total_games, 1),
                "games_per_second": self.total_games / max(self.total_time, 1e-6),
                "env_steps": self.total_env_steps, "capacity_replays": self.capacity_replays, "capacity_faults": self.capacity_faults,
                "depth_faults": self.depth_faults,
                "env_steps_per_second": self.total_env_steps / max(self.total_time, 1e-6)}

    def kernel_time(self):
        """(ms, launches) of the hot kernel over every engine this evaluator has created (HIP events on their streams)."""
        ms, n = 0.0, 0
        for eng in self._engines.values():
            if eng is not None:
                a, b = eng.kernel_time()
                ms, n = ms + a, n + b
        return ms, n

    def reset_stats(self):
        self.total_games = 0
        self.total_time = 0.0
        self.total_env_steps = 0
        self.total_decisions = 0
        for eng in self._engines.values():
            if eng is not None:
                eng.reset_stats()
