"""Oracle-backed stand-in for BatchEngine.rollout, for CPU tests of the host logic (Seam F,
schedule sharding, the gloo all-reduce).  Test infrastructure only."""
import numpy as np

import oracle_lib


THREADS = 1   # host threads of oracle_rollout_tier (oracle_rollout_fn_mt raises it)


def oracle_rollout_tier(weights, matches, deck_pairs, max_turns, tier, threads=None):
    """(counts, results, steps, faults) of the schedule on ONE build of the oracle (tier 0 standard, 1 extended, 2 large):
    orc_rollout_schedule, a loop over the matches on `threads` host threads inside the oracle library."""
    import ctypes
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    deck_pairs = np.ascontiguousarray(np.asarray(deck_pairs, dtype=np.uint8).reshape(-1, 2, 12))
    m = np.ascontiguousarray(matches, dtype=[("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])
    L = oracle_lib.lib(tier)
    L.orc_rollout_schedule.restype = ctypes.c_uint64
    L.orc_rollout_schedule.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    results = np.zeros(len(m), dtype=np.int8)
    steps = np.zeros(len(m), dtype=np.int32)
    faults = np.zeros(len(m), dtype=np.uint8)
    p = oracle_lib._p
    if len(m):
        L.orc_rollout_schedule(p(weights), p(m), len(m), p(deck_pairs), max_turns, threads or THREADS, p(results), p(steps), p(faults))
    counts = np.zeros((len(weights), 3), dtype=np.int64)
    np.add.at(counts[:, 0], m["p1"], results == 0)
    np.add.at(counts[:, 1], m["p1"], results == -1)
    np.add.at(counts[:, 2], m["p1"], 1)
    return counts, results, steps, faults


def oracle_rollout_fn(weights, matches, deck_pairs, max_turns, want_results=False, want_faults=False):
    """The product's rollout (monsoon_amd/fitness.py::_hip_rollout) on the CPU: every game on the smallest record its decks
    need, and the games that record cannot hold replayed on the next larger one (fitness.tiered_rollout)."""
    from monsoon_amd.fitness import tiered_rollout
    matches = np.asarray(matches)
    counts, results, steps, faults, _, _ = tiered_rollout(
        lambda tier, sub, sub_pairs: oracle_rollout_tier(weights, sub, sub_pairs, max_turns, tier), len(weights), matches, deck_pairs)
    if want_faults:
        return counts, results, steps, faults
    return (counts, results, steps) if want_results else counts


def oracle_rollout_fn_mt(weights, matches, deck_pairs, max_turns, want_results=False, threads=16, want_faults=False):
    """The same on several host threads: for the larger GPU-vs-CPU comparisons."""
    from monsoon_amd.fitness import tiered_rollout
    matches = np.asarray(matches)
    counts, results, steps, faults, _, _ = tiered_rollout(
        lambda tier, sub, sub_pairs: oracle_rollout_tier(weights, sub, sub_pairs, max_turns, tier, threads), len(weights), matches, deck_pairs)
    if want_faults:
        return counts, results, steps, faults
    return (counts, results, steps) if want_results else counts


def oracle_draw_decks(seeds, pool):
    """orc_draw_decks: the CPU restatement of monsoon_draw_decks (pinned to numpy's own draws by tests/golden/deck_draw_kat.npz)."""
    import ctypes
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    out = np.zeros((len(seeds), 2, 12), dtype=np.uint8)
    L = oracle_lib.lib()
    L.orc_draw_decks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    over = L.orc_draw_decks(oracle_lib._p(seeds), len(seeds), oracle_lib._p(pool), len(pool), oracle_lib._p(out))
    assert over == 0
    return out
