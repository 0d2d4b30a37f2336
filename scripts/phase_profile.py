#!/usr/bin/env python3
"""k_play phase timing (wave cycles per phase) with the profiling build libmonsoon_hip_prof.so.

    make -C monsoon_amd/csrc prof && gpurun -- python scripts/phase_profile.py [--games 65536]
The profiling library is a diagnostics build of the same source (-DMSB_PROF=1); the package never loads it.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import monsoon_amd._lib as L  # noqa: E402

L.LIB_PATH = os.path.join(REPO, "monsoon_amd", os.environ.get("MSB_PROF_LIB", "libmonsoon_hip_prof.so"))
from monsoon_amd.cards import deck_indices  # noqa: E402
from monsoon_amd.engine import BatchEngine  # noqa: E402

PHASES = ["stage", "legal mask", "before-features", "clone", "step", "after-features+score", "argmax+park", "commit+refill"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=65536)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--lanes", type=int, default=0)
    args = ap.parse_args()
    n = args.games
    eng = BatchEngine(n, lanes_per_game=args.lanes)
    deck = deck_indices("N12M")
    eng.reset(np.arange(n, dtype=np.uint32), np.stack([deck, deck]))
    eng.upload_weights(np.random.RandomState(2024).uniform(0, 1, 10).reshape(1, 10))
    eng.assign_players(np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32))
    for _ in range(args.warmup):
        eng.decide_round()
    eng.sync()
    eng.reset_stats()
    for _ in range(args.steps):
        eng.decide_round()
    eng.sync()
    c = np.zeros(192, dtype=np.uint64)
    eng._ck(eng.lib.monsoon_debug_counters(eng.h, c.ctypes.data_as(ctypes.c_void_p)), "counters")
    look, dec = int(c[0]), int(c[1])
    ph = c[8:16].astype(np.float64)
    tot = ph.sum()
    kms, launches = eng.kernel_time()
    print(f"games {n} decisions {dec} lookahead {look} ({look / max(dec, 1):.1f}/decision) k_play {kms / max(launches, 1):.3f} ms/launch")
    print(f"wave cycles per decision: {tot / max(dec, 1):.0f}")
    for name, v in zip(PHASES, ph):
        print(f"  {name:24s} {v / max(dec, 1):10.0f} cycles/decision  {100 * v / tot:5.1f} %")
    if c[96]:
        span, total, tail0, longest, nw, last_start = (float(c[i]) for i in range(96, 102))
        print(f"last launch (100 MHz wall clock): span {span / 100:.1f} us, {int(nw)} waves, mean wave {total / nw / 100:.1f} us, longest {longest / 100:.1f} us")
        print(f"  resident-wave average {total / span:.0f} of 4096; last wave started at {last_start / 100:.1f} us; 4096th-from-last wave ended at {tail0 / 100:.1f} us")
    print("function scopes (inclusive wave cycles of the sub-wave executing them; nested scopes count twice):")
    k = len(SCOPES)
    rows = sorted(zip(SCOPES, c[32:32 + k].astype(float), c[64:64 + k].astype(float), c[128:128 + k].astype(float), c[160:160 + k].astype(float)),
                  key=lambda r: -r[1])
    for name, cyc, calls, ent, ext in rows:
        if calls:
            extra = f"  entry {ent / calls:6.0f} exit {ext / calls:6.0f} cycles/call ({100 * (ent + ext) / tot:4.1f} % of all)" if ent or ext else ""
            print(f"  {name:16s} {cyc / dec:10.0f} cycles/decision {100 * cyc / tot:5.1f} %  {calls / dec:7.2f} calls/decision {cyc / calls:8.0f} cycles/call{extra}")


# round 3 (work-stack core): "run() loop" = the whole dispatcher loop incl. handlers; "move" / "run_ability" = the F_MOVE / F_RUNAB
# handlers; "F_EACH" / "F_TURN" / "next_turn prefix" the out-of-line ones
SCOPES = ["step", "player_play", "new_entity", "run_ability", "ability_entity", "ability_spell", "get_targets", "shape_tiles",
          "shape_targets", "deal_damage", "destroy", "front_line", "set_path", "move", "run() loop", "F_EACH", "draw",
          "next_turn prefix", "F_TURN", "legal_mask", "shuffle", "sorted_head", "spawn", "respawn", "teleport", "push_pull", "empty_front",
          "begin_step", "obs_raises", "features"]


if __name__ == "__main__":
    main()
