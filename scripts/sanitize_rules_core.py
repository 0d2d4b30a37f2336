#!/usr/bin/env python3
"""The rules core under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on this pool): the
host build of rules.h / abilities (the oracle, test infrastructure) plays heuristic rollouts on named and random decks,
standard and extended record.  Build + run:
    bash scripts/sanitize_rules_core.sh
"""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.LIB_PATH = '/tmp/liboracle_asan.so'
oracle_lib.LIB_PATH_EXT = '/tmp/liboracle_ext_asan.so'
oracle_lib.LIB_PATH_BIG = '/tmp/liboracle_big_asan.so'
import numpy as np
from monsoon_amd.cards import CARD_IDS, CARD_INDEX, supported_pool, deck_indices
W0=np.random.RandomState(2024).uniform(0,1,10)
# heuristic rollouts on named decks and random decks, both builds
for ext in (False, True, 2):
    pool=np.array([i for i,c in enumerate(CARD_IDS) if c not in ("up01","up02","up03") and (ext or c not in ("ua20","b005"))],dtype=np.uint8)
    orc=oracle_lib.Oracle(1, extended=ext)
    n=0
    for g in range(160):
        rs=np.random.RandomState(g ^ 0x9E3779B9)
        a,b=rs.choice(pool,12,replace=False),rs.choice(pool,12,replace=False)
        orc.reset(0,30000+g,a,b); r=orc.rollout(0,W0,W0,200); n+=r['lookahead']
    for name in ("N12M","N12V","S12"):
        d=deck_indices(name)
        for g in range(6):
            orc.reset(0,g,d,d); r=orc.rollout(0,W0,W0,200); n+=r['lookahead']
    # include up0x decks (observation raises)
    d=deck_indices("N12M").copy(); d[0]=CARD_INDEX["up01"]
    orc.reset(0,1,d,d); orc.rollout(0,W0,W0,50)
    if ext:   # games of the C5 family whose nested b005 memories overflow the extended record (capacity paths, worlds, GC)
        from c5_games import C5_OVERFLOWING, c5_games
        m, pairs = c5_games(C5_OVERFLOWING)
        for k in range(len(C5_OVERFLOWING)):
            orc.reset(0, int(m["seed"][k]), pairs[k, 0], pairs[k, 1]); r = orc.rollout(0, W0, W0, 200); n += r['lookahead']
    print({False: "std", True: "ext", 2: "big"}[ext], "look-ahead steps under ASan/UBSan:", n, flush=True)
print("sanitizers clean (rollouts)")
# the scenario fixtures (the reference's own tests, call by call): state builder + single engine calls, both builds
import scenario_lib as S
ext_cards = [CARD_INDEX["ua20"], CARD_INDEX["b005"]]
orcs = {False: oracle_lib.Oracle(1), True: oracle_lib.Oracle(1, extended=True)}
n = 0
for case in S.load():
    for rec in case["records"]:
        for ext in sorted({S.needs_extended(rec, ext_cards), True}):   # everything also on the extended build
            st = rec["before"]
            orcs[ext].scn_build(0, st["seed"], st["stream_pos"], S.encode_state(st))
            orcs[ext].scn_op(0, S.encode_op(rec))
            n += 1
print("scenario calls under ASan/UBSan:", n)
print("sanitizers clean (scenarios)")
