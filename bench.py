#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the heuristic self-play hot path at 65 536 concurrent games.

Workload c2 (default; BASELINE.json configs[1]): 65 536 games per GPU, seeds
rank*65536 + 0..65535, deck N12M vs N12M, both players the weight vector
W0 = RandomState(2024).uniform(0,1,10), synthetic inputs generated on the device.  Weak scaling.
One bench "step" = one pass of the hot path over the batch: every live game advances by --rounds decisions (default
8) in one launch; a decision = HeuristicAgent.select_action (1-ply look-ahead over all legal actions, feature
delta, score, argmax) + committing the chosen successor.  A game's record stays in LDS from its first to its last
decision of the step; --rounds 1 is one decision round per launch (the round-1 form of this bench).

Workloads c3 / c4 / c5 (--workload ...; BASELINE.json configs[2..4]): one generation of the GA's evaluation
through FitnessEvaluator.evaluate_population per step -- c3: 256 individuals x 64 games on N12M; c4: population 1 024,
64 games per individual on the Swarm deck S12; c5: population 4 096 x 128 games, per-game decks drawn on the device from
the 109 observable cards, every game on the smallest record its decks need.  The schedule is sharded by row individual
over the ranks and the per-individual counters are summed with one all-reduce (RCCL).  Strong scaling: the games of a
generation are fixed; host time (schedule, decks, upload, collection) is inside the timed region.

`value` counts the Stormbound.step transitions actually EXECUTED in the timed region (the look-ahead steps; the
committed successor is one of them and is not re-executed) divided by the wall time of the region, summed over
ranks / max over ranks.  Inputs are resident in HBM when the timed region starts (c2; c4 uploads 80 KB of weights
and a 1 MB schedule per generation, as the GA driver does).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|rollout] [--deck D] [--games G] [--lanes U] [--rounds R] [--no-cpu]
With --gpus N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself (one process per
GPU through torch.distributed.run, before anything here touches a GPU) and relays rank 0's line; under
torch.distributed.run it is one of the ranks.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

W0 = np.random.RandomState(2024).uniform(0, 1, 10)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per env-step = 904 per look-ahead (read parent record + RNG
# words + write score) + 1736 per committed decision (+ write the successor record)
BYTES_LOOKAHEAD, BYTES_COMMIT = 904, 1736
# the Python reference timed in the BUILD CONTAINER (one Xeon 2.1 GHz core), not on the GPU box
REFERENCE_PYTHON = {"value": 315.0, "range": [280.0, 350.0], "unit": "env-steps/s", "cores": 1,
                    "where": "build container (BASELINE.md §2), not this box",
                    "sample": "2 N12M heuristic self-play games to 200 decisions, look-ahead steps included"}
METRIC = "env-steps/sec (whole node) at 65536 concurrent games; bit-exact vs CPU replay"


def recorded_traffic():
    """HBM bytes per k_play launch and the kernel's real ceilings (lanes per VALU instruction, VALU busy, wavefronts per
    CU, scratch per lane) from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    scripts/summarize_profile.py) with the configuration they were measured on -- OFFLINE numbers, labelled so."""
    p = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(p):
        d = json.load(open(p))
        return d.get("bytes_per_launch"), d.get("measured_on", "profiles/traffic.json (separate rocprofv3 --pmc runs)"), d.get("ceilings")
    return None, None, None


def cpu_baseline(sample_games, max_turns, threads):
    """The CPU replay oracle (oracle/, kind 'port') timed on the host cores: a bounded sample of the
    same workload (same deck, weights, seeds 0..sample-1, same step accounting)."""
    from monsoon_amd.cards import deck_indices
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
    # "1 thread and all host cores" -- the same replay on one thread, a 32nd of the sample
    one = max(64, sample_games // 32)
    for i in range(one):
        orc.reset(i, i, deck, deck)
    t1 = time.perf_counter()
    total1, _, _, _ = orc.rollout_batch(one, W0, max_turns, 1)
    dt1 = time.perf_counter() - t1
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{sample_games} N12M self-play games to {max_turns} decisions ({total} look-ahead steps, "
                      f"{dt:.2f} s wall on {cores} threads)",
            "one_thread": {"value": total1 / dt1, "unit": "env-steps/s", "cores": 1,
                           "sample": f"{one} of those games ({total1} look-ahead steps, {dt1:.2f} s)"},
            "reference_python": REFERENCE_PYTHON}


class FakeEngine:
    """TEST HOOK (MONSOON_BENCH_FAKE=1, tests/test_distributed_cpu.py only): stands in for the HIP engine so that the
    launcher, the rank plumbing and the all-reduce of this file run on CPU ranks with the gloo backend.  It plays no
    game and its numbers mean nothing; bench.py never selects it by itself."""

    def __init__(self, n):
        self.n, self.rounds = n, 0

    def variant(self):
        return (0, 0)

    def play_rounds(self, rounds):
        self.rounds += rounds

    def reset(self, seeds, decks):
        self.seeds = seeds

    def upload_weights(self, w): pass
    def assign_players(self, a, b): pass
    def sync(self): pass

    def decide_round(self):
        self.rounds += 1

    def reset_stats(self):
        self.rounds = 0

    def stats(self):
        return {"lookahead_steps": 17 * self.n * self.rounds, "decisions": self.n * self.rounds, "games_finished": 0,
                "faults": 0, "capacity_faults": 0, "lookahead_capacity_faults": 0}

    def kernel_time(self):
        return 1.0 * self.rounds, self.rounds


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5", "rollout"], default="c2")
    ap.add_argument("--deck", default="N12M", help="rollout workload: a named deck (both sides), or 'random' = C5's per-game 12-card decks")
    ap.add_argument("--games", type=int, default=65536)
    ap.add_argument("--lanes", type=int, default=0, help="candidate lanes per game (0 = build default)")
    ap.add_argument("--rounds", type=int, default=8, help="c2: decisions every game advances per step (one launch)")
    ap.add_argument("--stack", type=int, default=0, help="per-lane scratch stack bytes (0 = library default)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32768)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 12 if args.workload == "c2" else (2 if args.workload == "c5" else 5)
    if args.warmup is None:
        args.warmup = 3 if args.workload == "c2" else 1
    return args


def launch_ranks(args, argv):
    """--gpus N > 1 without a rendezvous in the environment: start the N ranks (one process per GPU) as a CHILD job
    and exit with its code.  Nothing in this process has touched a GPU (torch.cuda.device_count() does not)."""
    fake = os.environ.get("MONSOON_BENCH_FAKE") == "1"
    if not fake:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} asked for, this machine shows {have} GPU(s); refusing to report a "
                             f"{args.gpus}-GPU number from fewer devices\n")
            return 2
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fake = os.environ.get("MONSOON_BENCH_FAKE") == "1"
    backend = os.environ.get("MONSOON_BENCH_BACKEND", "nccl")   # gloo only with the fake engine (CPU test)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    dist = None
    dev = "cpu"
    if world > 1 or os.environ.get("MONSOON_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                raise SystemExit(f"bench.py: rank {rank} has no GPU {local_rank} (device_count {torch.cuda.device_count()})")
            torch.cuda.set_device(local_rank)
            dev = "cuda"
        dist.init_process_group(backend)   # "nccl" IS RCCL on ROCm

    def barrier():
        if dist is not None:
            dist.barrier()
            if dev == "cuda":
                import torch
                torch.cuda.synchronize()

    def all_sum(vals, dtype="int64"):
        if dist is None:
            return list(vals)
        import torch
        t = torch.tensor(list(vals), dtype=getattr(torch, dtype), device=dev)
        dist.all_reduce(t)
        return t.tolist()

    def all_max(v):
        if dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    if args.workload == "c2":
        line = bench_c2(args, rank, world, local_rank, fake, barrier, all_sum, all_max)
    elif args.workload == "rollout":
        line = bench_rollout(args, rank, world, local_rank, barrier, all_sum, all_max)
    else:
        line = bench_ga(args, rank, world, local_rank, barrier, all_sum, all_max)
    # ranks that really took part in the collective (the job's size as RCCL saw it)
    took_part = int(all_sum([1])[0])
    if rank == 0:
        line["n_gpus"] = took_part
        line["ranks"] = {"launched": world, "in_all_reduce": took_part, "backend": backend if dist is not None else None}
        if took_part != args.gpus:
            raise SystemExit(f"bench.py: {took_part} rank(s) completed the all-reduce, --gpus was {args.gpus}")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_c2(args, rank, world, local_rank, fake, barrier, all_sum, all_max):
    from monsoon_amd.cards import deck_indices
    n = args.games
    if fake:
        eng = FakeEngine(n)
    else:
        from monsoon_amd.engine import BatchEngine
        eng = BatchEngine(n, device=local_rank, lanes_per_game=args.lanes, stack_bytes=args.stack)
    try:
        var_u, var_w = eng.variant()
    except AttributeError:   # an older build loaded through MONSOON_LIB (A/B runs)
        var_u, var_w = args.lanes or 8, 0
    play = (lambda: eng.play_rounds(args.rounds)) if (args.rounds > 1 and hasattr(eng, "play_rounds")) else eng.decide_round
    deck = deck_indices("N12M")
    seeds = (np.arange(n, dtype=np.uint64) + np.uint64(rank) * np.uint64(n)).astype(np.uint32)
    eng.reset(seeds, np.stack([deck, deck]))
    eng.upload_weights(W0.reshape(1, 10))
    eng.assign_players(np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32))
    for _ in range(args.warmup):
        play()
    eng.sync()
    eng.reset_stats()

    barrier()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        play()
    eng.sync()
    # the path's one real exchange: per-individual {wins, draws, games} summed over ranks (RCCL)
    st_now = eng.stats()
    all_sum([st_now["games_finished"], 0, n])
    barrier()
    eng.sync()
    dt = time.perf_counter() - t0

    st = eng.stats()
    kms, launches = eng.kernel_time()
    look, dec = st["lookahead_steps"], st["decisions"]
    tot_look, tot_dec = all_sum([look, dec])
    max_dt = all_max(dt)
    avg_launch_s = (kms / 1000.0) / max(launches, 1)
    alg_bytes = (BYTES_LOOKAHEAD * look + BYTES_COMMIT * dec) / max(launches, 1)
    achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic, traffic_on, ceilings = recorded_traffic()
    # what the kernel has to move at the least: every game's record and meta row in and out once per launch (the RNG
    # words it draws and the block refills come on top; SURVEY's per-step figure assumes a record read per look-ahead)
    physical = float(n) * 2 * (752 + 32)
    line = {
        "metric": METRIC,
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
        "config": {"workload": f"C2: 65536 N12M-vs-N12M heuristic self-play games per GPU, W0 both sides, every game advances "
                               f"{args.rounds} decision(s) per step (one launch)", "games_per_gpu": n, "rounds_per_step": args.rounds,
                   "lanes_per_game": var_u, "waves_per_simd": var_w,
                   "parallelism": f"games sharded x{world}, no data-path collective"},
        "decisions_per_s": tot_dec / max_dt,
        "lookahead_per_decision": tot_look / max(tot_dec, 1),
        "faults": st["faults"], "capacity_faults": st["capacity_faults"],
        "lookahead_capacity_faults": st.get("lookahead_capacity_faults", 0),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_measured_on": traffic_on,
                     "kernel": "k_play", "avg_launch_ms": 1000.0 * avg_launch_s, "launches": launches,
                     "algorithmic_bytes_per_launch": alg_bytes, "physical_bytes_per_launch": physical},
    }
    if ceilings:   # offline, from the same rocprofv3 passes as `traffic`: the ceilings that bind this integer, LDS-resident kernel
        line["roofline"].update(ceilings)
    if not args.no_cpu and world == 1 and rank == 0 and not fake:
        line["cpu_baseline"] = cpu_baseline(args.cpu_sample, 200, args.cpu_threads)
    return line


def bench_rollout(args, rank, world, local_rank, barrier, all_sum, all_max):
    """Whole games: one step = --games fresh games (reset + monsoon_rollout to a winner, a fault or 200 decisions), on a
    named deck or on C5's per-game random decks (109 observable cards, extended-record build).  For profiling the
    configurations other than C2 (S12: short, divergent games; random decks: worst-case card divergence)."""
    from monsoon_amd.cards import CARD_IDS, deck_indices
    from monsoon_amd.engine import BatchEngine
    n = args.games
    random_decks = args.deck == "random"
    eng = BatchEngine(n, device=local_rank, lanes_per_game=args.lanes, stack_bytes=args.stack, extended=random_decks)
    if random_decks:
        pool = np.array([i for i, c in enumerate(CARD_IDS) if c not in ("up01", "up02", "up03")], dtype=np.uint8)
        pairs = np.zeros((n, 2, 12), dtype=np.uint8)
        for g in range(n):
            rs = np.random.RandomState((g + rank * n) ^ 0x9E3779B9)
            pairs[g, 0], pairs[g, 1] = rs.choice(pool, 12, replace=False), rs.choice(pool, 12, replace=False)
    else:
        deck = deck_indices(args.deck)
        pairs = np.stack([deck, deck])[None]
    w = np.random.RandomState(2024).uniform(0, 1, (2, 10))

    def one(step):
        m = np.zeros(n, dtype=[("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])
        m["p2"] = 1
        m["seed"] = np.arange(n) + (step * world + rank) * n
        if random_decks:
            m["deck"] = np.arange(n)
        return eng.rollout(w, m, pairs, 200, want_results=True)

    for k in range(args.warmup):
        one(k)
    eng.reset_stats()
    barrier()
    t0 = time.perf_counter()
    decisions = 0
    for k in range(args.steps):
        _, _, steps = one(args.warmup + k)
        decisions += int(steps.sum())
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    kms, launches = eng.kernel_time()
    tot_look, tot_dec = all_sum([st["lookahead_steps"], decisions])
    max_dt = all_max(dt)
    return {
        "metric": METRIC, "value": tot_look / max_dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * max_dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/i16 state, f64 draw+score", "data": "synthetic",
        "config": {"workload": f"rollout: {n} fresh games per step on deck {args.deck}, played to the end (reset + upload + one k_play launch + "
                               f"collect)", "games_per_gpu": n, "lanes_per_game": eng.variant()[0], "extended_record": random_decks},
        "decisions_per_s": tot_dec / max_dt, "lookahead_per_decision": tot_look / max(tot_dec, 1),
        "mean_game_length": decisions / (n * args.steps), "faults": st["faults"], "capacity_faults": st["capacity_faults"],
        "lookahead_capacity_faults": st["lookahead_capacity_faults"],
        "roofline": {"bound": "hbm", "achieved": (BYTES_LOOKAHEAD * st["lookahead_steps"] + BYTES_COMMIT * decisions) / max(kms / 1000.0, 1e-9) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "kernel": "k_play", "avg_launch_ms": kms / max(launches, 1),
                     "launches": launches},
    }


GA_WORKLOADS = {
    # name: (individuals evaluated per generation, games per individual, deck, what BASELINE.json calls it)
    "c3": (256, 64, "N12M", "C3: mu = lambda = 128 (256 evaluated individuals) x 64 ring games per generation (16384 games), N12M deck both sides"),
    "c4": (1024, 64, "S12", "C4: population 1024 x 64 ring games per generation (65536 games), S12 deck both sides"),
    "c5": (4096, 128, "random109", "C5: population 4096 x 128 ring games per generation (524288 games), per-game decks drawn on the device "
                                   "from the 109 observable cards, every game on the smallest record its decks need"),
}


def bench_ga(args, rank, world, local_rank, barrier, all_sum, all_max):
    """Strong scaling: one step = one generation of the named configuration through FitnessEvaluator.evaluate_population --
    schedule, per-game decks (c5: monsoon_draw_decks), upload, rollouts, collection and the all-reduce of the counters are
    all inside the timed region; the schedule is sharded by row individual over the ranks."""
    from monsoon_amd.config import EvolutionaryConfig
    from monsoon_amd.fitness import FitnessEvaluator
    from monsoon_amd.weights import WeightVector
    n_ind, gpi, deck, what = GA_WORKLOADS[args.workload]
    cfg = EvolutionaryConfig(mu=n_ind, lambda_=n_ind, schedule="ring", games_per_individual=gpi, deck=deck, max_turns=200,
                             max_concurrent_games=65536, lanes_per_game=args.lanes,
                             concurrent_tiers=os.environ.get("MONSOON_CONCURRENT_TIERS", "1") != "0")   # development knob (A/B)
    np.random.seed(42)
    pop = [WeightVector(10) for _ in range(n_ind)]
    ev = FitnessEvaluator(cfg, device=local_rank)
    ev.use_hall_of_fame = False
    for g in range(args.warmup):
        ev.evaluate_population(pop, generation=g)
    ev.reset_stats()
    ev.tier_games, ev.capacity_replays, ev.capacity_faults = [0, 0], 0, 0
    barrier()
    t0 = time.perf_counter()
    for g in range(args.steps):
        fit = ev.evaluate_population(pop, generation=100 + g)   # includes the all-reduce of the counters
    barrier()
    dt = time.perf_counter() - t0
    kms, launches = ev.kernel_time()
    tot_steps, tot_dec = all_sum([ev.total_env_steps, ev.total_decisions])
    max_dt = all_max(dt)
    alg = BYTES_LOOKAHEAD * ev.total_env_steps + BYTES_COMMIT * ev.total_decisions
    achieved = alg / max(kms / 1000.0, 1e-9) / 1e9
    return {
        "metric": METRIC,
        "value": tot_steps / max_dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * max_dt / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8/i16 state, f64 draw+score",
        "data": "synthetic",
        "config": {"workload": what + ", FitnessEvaluator.evaluate_population, schedule sharded by row individual",
                   "games_per_step": n_ind * gpi, "parallelism": f"row individuals sharded x{world}, one all-reduce of int64[{n_ind}][3]"},
        "games_per_s": n_ind * gpi * args.steps / max_dt,
        "decisions_per_s": tot_dec / max_dt,
        "mean_fitness": float(np.mean(fit)),
        "record_tiers": {"standard": ev.tier_games[0], "extended": ev.tier_games[1], "replayed_on_a_larger_record": ev.capacity_replays,
                         "left_on_a_record_limit": ev.capacity_faults},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_play (rank 0, all record tiers)", "kernel_ms_per_step": kms / max(args.steps, 1), "launches": launches,
                     # kernel time is summed over the handles of the record tiers, whose launches overlap when a schedule
                     # has games on two records (config.concurrent_tiers): the difference is only meaningful for one tier
                     "host_and_other_ms_per_step": (1000.0 * dt / args.steps - kms / max(args.steps, 1)) if ev.tier_games[1] == 0 or not cfg.concurrent_tiers else None},
    }


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
