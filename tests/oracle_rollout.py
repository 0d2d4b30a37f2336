"""Oracle-backed stand-in for BatchEngine.rollout, for CPU tests of the host logic (Seam F,
schedule sharding, the gloo all-reduce).  Test infrastructure only."""
import numpy as np

import oracle_lib


def oracle_rollout_fn(weights, matches, deck_pairs, max_turns, want_results=False):
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    deck_pairs = np.asarray(deck_pairs, dtype=np.uint8).reshape(-1, 2, 12)
    counts = np.zeros((len(weights), 3), dtype=np.int64)
    from monsoon_amd.cards import needs_extended
    orc = oracle_lib.Oracle(1, extended=bool(needs_extended(deck_pairs)))   # ua20 / b005: extended record, like the product
    results = np.zeros(len(matches), dtype=np.int8)
    steps = np.zeros(len(matches), dtype=np.int32)
    for k, m in enumerate(matches):
        d = deck_pairs[int(m["deck"])]
        orc.reset(0, int(m["seed"]), d[0], d[1])
        r = orc.rollout(0, weights[int(m["p1"])], weights[int(m["p2"])], max_turns)
        results[k], steps[k] = r["result"], r["steps"]
        if r["result"] == 0:
            counts[int(m["p1"]), 0] += 1
        elif r["result"] == -1:
            counts[int(m["p1"]), 1] += 1
        counts[int(m["p1"]), 2] += 1
    return (counts, results, steps) if want_results else counts


def oracle_rollout_fn_mt(weights, matches, deck_pairs, max_turns, want_results=False, threads=16):
    """The same on several host threads (ctypes releases the GIL): for the larger GPU-vs-CPU comparisons."""
    from concurrent.futures import ThreadPoolExecutor
    matches = np.asarray(matches)
    chunks = [c for c in np.array_split(np.arange(len(matches)), threads) if len(c)]
    with ThreadPoolExecutor(len(chunks)) as ex:
        parts = list(ex.map(lambda idx: oracle_rollout_fn(weights, matches[idx], deck_pairs, max_turns, want_results=True), chunks))
    counts = sum(p[0] for p in parts)
    results = np.concatenate([p[1] for p in parts])
    steps = np.concatenate([p[2] for p in parts])
    return (counts, results, steps) if want_results else counts
