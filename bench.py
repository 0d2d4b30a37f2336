#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the heuristic self-play hot path at 65 536 concurrent games.

Workload (BASELINE.json configs[1]): 65 536 games per GPU, seeds
rank*65536 + 0..65535, deck N12M vs N12M, both players the weight vector
W0 = RandomState(2024).uniform(0,1,10), synthetic inputs generated on the device.

One bench "step" = one decision round: every live game runs HeuristicAgent.select_action (1-ply
look-ahead over all legal actions, feature delta, score, argmax) and commits the chosen successor.
`value` counts the Stormbound.step transitions actually EXECUTED in the timed region (the
look-ahead steps; the committed successor is one of them and is not re-executed) divided by the
wall time of the region, summed over ranks / max over ranks.  Inputs are resident in HBM when the
timed region starts; nothing crosses PCIe inside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--lanes U] [--no-cpu]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

W0 = np.random.RandomState(2024).uniform(0, 1, 10)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per env-step = 904 per look-ahead (read parent record + RNG
# words + write score) + 1736 per committed decision (+ write the successor record)
BYTES_LOOKAHEAD, BYTES_COMMIT = 904, 1736


def recorded_traffic():
    """HBM bytes per k_decide launch from the committed rocprofv3 PMC passes (profiles/traffic.json,
    written by scripts/summarize_profile.py); None if no profile has been recorded."""
    p = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(p):
        return json.load(open(p)).get("bytes_per_launch")
    return None


def cpu_baseline(sample_games, max_turns, threads):
    """The CPU replay oracle (oracle/, kind 'port') timed on the host cores: a bounded sample of the
    same workload (same deck, weights, seeds 0..sample-1, same step accounting)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, threads))   # a one-GPU box's CPU share is 16 cores
    deck = deck_indices("N12M")
    orc = oracle_lib.Oracle(sample_games)
    for i in range(sample_games):
        orc.reset(i, i, deck, deck)
    t0 = time.perf_counter()
    total, _, steps, _ = orc.rollout_batch(sample_games, W0, max_turns, cores)
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{sample_games} N12M self-play games to {max_turns} decisions ({total} look-ahead steps, "
                      f"{dt:.2f} s wall on {cores} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--games", type=int, default=65536)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--stack", type=int, default=0, help="per-lane scratch stack bytes (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("MONSOON_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")   # RCCL on ROCm

    n = args.games
    eng = BatchEngine(n, device=local_rank, lanes_per_game=args.lanes, stack_bytes=args.stack)
    deck = deck_indices("N12M")
    seeds = (np.arange(n, dtype=np.uint64) + np.uint64(rank) * np.uint64(n)).astype(np.uint32)
    eng.reset(seeds, np.stack([deck, deck]))
    eng.upload_weights(W0.reshape(1, 10))
    eng.assign_players(np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32))

    for _ in range(args.warmup):
        eng.decide_round()
    eng.sync()
    eng.reset_stats()

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.decide_round()
    eng.sync()
    if dist is not None:
        # the path's one real exchange: per-individual {wins, draws, games} summed over ranks (RCCL)
        import torch
        st_now = eng.stats()
        counts = torch.tensor([[st_now["games_finished"], 0, n]], dtype=torch.int64, device="cuda")
        dist.all_reduce(counts)
        torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0

    st = eng.stats()
    kms, launches = eng.kernel_time()
    look, dec = st["lookahead_steps"], st["decisions"]
    tot_look, tot_dec, max_dt = look, dec, dt
    if dist is not None:
        import torch
        t = torch.tensor([look, dec], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        tm = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tot_look, tot_dec, max_dt = int(t[0]), int(t[1]), float(tm[0])

    if rank == 0:
        avg_launch_s = (kms / 1000.0) / max(launches, 1)
        alg_bytes = (BYTES_LOOKAHEAD * look + BYTES_COMMIT * dec) / max(launches, 1)
        achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        line = {
            "metric": "env-steps/sec (whole node) at 65536 concurrent games; bit-exact vs CPU replay",
            "value": tot_look / max_dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * max_dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i16 state, f64 draw+score",
            "data": "synthetic",
            "config": {"workload": "C2: 65536 N12M-vs-N12M heuristic self-play games per GPU, W0 both sides, one decision "
                                   "round per step", "games_per_gpu": n, "lanes_per_game": args.lanes or 8,
                       "parallelism": f"games sharded x{world}, no data-path collective"},
            "decisions_per_s": tot_dec / max_dt,
            "lookahead_per_decision": tot_look / max(tot_dec, 1),
            "faults": st["faults"], "capacity_faults": st["capacity_faults"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": recorded_traffic(),
                         "kernel": "k_decide", "avg_launch_ms": 1000.0 * avg_launch_s, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, 200, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
