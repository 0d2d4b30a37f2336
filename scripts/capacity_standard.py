#!/usr/bin/env python3
"""How many entity slots does the STANDARD record need?  Heuristic rollouts of five deck families on a host build of the
oracle with -DMSB_CAP_ENT=<slots> (CPU only, test infrastructure); prints the fault codes per family: code 16 = a step
wanted one slot more than the record has.

    for E in 24 22 21; do g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I monsoon_amd/csrc -DMSB_CAP_ENT=$E \\
        -shared -o oracle/_cap/liboracle_e$E.so oracle/oracle.cpp -lpthread; python scripts/capacity_standard.py oracle/_cap/liboracle_e$E.so; done
Round 3: 24 slots {16: 0}, 22 slots {16: 1}, 21 slots {16: 30} of 68 000 games -> the record holds 24."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, REPO)
import oracle_lib
from oracle_rollout import oracle_rollout_tier
from monsoon_amd.cards import deck_indices, CARD_IDS, DECKS
lib = sys.argv[1]
oracle_lib._libs[(0,"oracle")] = oracle_lib.lib(lib, core="x")
rs = np.random.RandomState(5)
W = rs.uniform(0,1,(8,10)); W[0] = np.random.RandomState(2024).uniform(0,1,10)
dt=[("p1","<i4"),("p2","<i4"),("seed","<u4"),("deck","<u4")]
def run(name, pairs, n):
    m = np.zeros(n, dtype=dt); m["seed"] = rs.randint(0,2**31,n); m["p1"]=rs.randint(0,8,n); m["p2"]=rs.randint(0,8,n)
    m["deck"] = np.arange(n) % len(pairs)
    t=time.time(); c,r,s,f = oracle_rollout_tier(W, m, pairs, 200, 0, threads=8)
    codes,cnt=np.unique(f,return_counts=True)
    print(name, n, "games", f"{time.time()-t:.1f}s", dict(zip(codes.tolist(),cnt.tolist())), flush=True)
for d in ("N12M","S12","N12V"):
    dk = deck_indices(d); run(d, np.stack([dk,dk])[None], 12000)
run("IRONCLAD/SWARM", np.stack([deck_indices("IRONCLAD"), deck_indices("SWARM")])[None], 12000)
pool = np.array([i for i,c in enumerate(CARD_IDS) if c not in ("up01","up02","up03","ua20","b005")], dtype=np.uint8)
pairs = np.stack([np.stack([rs.choice(pool,12,replace=False), rs.choice(pool,12,replace=False)]) for _ in range(20000)])
run("random107", pairs, 20000)
