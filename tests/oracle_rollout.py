"""Oracle-backed stand-in for BatchEngine.rollout, for CPU tests of the host logic (Seam F,
schedule sharding, the gloo all-reduce).  Test infrastructure only."""
import numpy as np

import oracle_lib


def oracle_rollout_tier(weights, matches, deck_pairs, max_turns, tier):
    """(counts, results, steps, faults) of the schedule on ONE build of the oracle (tier 0 standard, 1 extended, 2 large)."""
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    deck_pairs = np.asarray(deck_pairs, dtype=np.uint8).reshape(-1, 2, 12)
    counts = np.zeros((len(weights), 3), dtype=np.int64)
    orc = oracle_lib.Oracle(1, extended=tier)
    results = np.zeros(len(matches), dtype=np.int8)
    steps = np.zeros(len(matches), dtype=np.int32)
    faults = np.zeros(len(matches), dtype=np.uint8)
    for k, m in enumerate(matches):
        d = deck_pairs[int(m["deck"])]
        orc.reset(0, int(m["seed"]), d[0], d[1])
        r = orc.rollout(0, weights[int(m["p1"])], weights[int(m["p2"])], max_turns)
        results[k], steps[k], faults[k] = r["result"], r["steps"], orc.game_fault(0)
        if r["result"] == 0:
            counts[int(m["p1"]), 0] += 1
        elif r["result"] == -1:
            counts[int(m["p1"]), 1] += 1
        counts[int(m["p1"]), 2] += 1
    return counts, results, steps, faults


def oracle_rollout_fn(weights, matches, deck_pairs, max_turns, want_results=False, want_faults=False):
    """The product's rollout (monsoon_amd/fitness.py::_hip_rollout) on the CPU: the extended record for decks holding
    ua20 / b005, and the games that record cannot hold replayed on the large one."""
    from monsoon_amd.cards import needs_extended
    from monsoon_amd.fitness import replace_capacity_faulted
    matches = np.asarray(matches)
    ext = int(bool(needs_extended(np.asarray(deck_pairs, dtype=np.uint8).reshape(-1, 2, 12))))
    counts, results, steps, faults = oracle_rollout_tier(weights, matches, deck_pairs, max_turns, ext)
    for tier in range(ext + 1, 3):
        replace_capacity_faulted(counts, results, steps, faults, matches, lambda sub, t=tier: oracle_rollout_tier(weights, sub, deck_pairs, max_turns, t))
    if want_faults:
        return counts, results, steps, faults
    return (counts, results, steps) if want_results else counts


def oracle_rollout_fn_mt(weights, matches, deck_pairs, max_turns, want_results=False, threads=16, want_faults=False):
    """The same on several host threads (ctypes releases the GIL): for the larger GPU-vs-CPU comparisons."""
    from concurrent.futures import ThreadPoolExecutor
    matches = np.asarray(matches)
    chunks = [c for c in np.array_split(np.arange(len(matches)), threads) if len(c)]
    with ThreadPoolExecutor(len(chunks)) as ex:
        parts = list(ex.map(lambda idx: oracle_rollout_fn(weights, matches[idx], deck_pairs, max_turns, want_faults=True), chunks))
    counts = sum(p[0] for p in parts)
    results = np.concatenate([p[1] for p in parts])
    steps = np.concatenate([p[2] for p in parts])
    if want_faults:
        return counts, results, steps, np.concatenate([p[3] for p in parts])
    return (counts, results, steps) if want_results else counts
