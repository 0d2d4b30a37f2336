"""CPU tests: the oracle (host build of the rules core) against the golden vectors generated from
the Python reference (oracle/pyref/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import oracle_lib


@pytest.fixture(autouse=True, params=["oracle", "product"])
def rules_core(request, monkeypatch):
    """Every test of this module runs twice: over the ORACLE (oracle/recursive/: the reference's call structure kept) and over
    the host build of the PRODUCT's rules core (monsoon_amd/csrc/rules.h: the explicit work stack the HIP kernels run).  Both
    must reproduce the reference's fixtures; oracle_lib.Oracle picks the library by this environment variable."""
    monkeypatch.setenv("MSB_ORACLE_CORE", request.param)
    return request.param



def _p(a):
    return a.ctypes.data_as(__import__("ctypes").c_void_p)


def test_rng_known_answers(oracle_mod, gold):
    import ctypes
    L = oracle_mod.lib()
    r = gold("rng_kat.npz")
    b = np.ascontiguousarray(r["randint_bounds"])
    for s in r["seeds"]:
        s = int(s)
        out = np.zeros(1500, dtype=np.uint32)
        L.orc_rng_u32(ctypes.c_uint32(s), 1500, _p(out))
        assert np.array_equal(out, r[f"u32_{s}"])
        d = np.zeros(400)
        L.orc_rng_random(ctypes.c_uint32(s), 400, _p(d))
        assert np.array_equal(d.view(np.uint64), r[f"random_{s}"].view(np.uint64))
        o = np.zeros(len(b), dtype=np.int32)
        L.orc_rng_randint(ctypes.c_uint32(s), len(b), _p(b), _p(o))
        assert np.array_equal(o, r[f"randint_{s}"])
        sh = np.zeros((20, 12), dtype=np.int32)
        L.orc_rng_shuffle(ctypes.c_uint32(s), 12, 20, _p(sh))
        assert np.array_equal(sh, r[f"shuffle12_{s}"])


def test_score_known_answers(oracle_mod, gold):
    """np.dot(w, delta) for n=10 == sequential FMA chain, bit for bit."""
    L = oracle_mod.lib()
    k = gold("score_kat.npz")
    w, b, a, sc = (np.ascontiguousarray(k[n]) for n in ("w", "before", "after", "score"))
    for i in range(len(sc)):
        s = L.orc_score(_p(w[i]), _p(b[i]), _p(a[i]))
        assert np.float64(s).view(np.uint64) == sc[i].view(np.uint64), i


def test_initial_states(oracle_mod, gold):
    g = gold("initial_states.npz")
    orc = oracle_mod.Oracle(1)
    for k in range(len(g["seeds"])):
        deck = g["decks"][k]
        f = orc.reset(0, int(g["seeds"][k]), deck, deck)
        assert f == 0
        assert orc.canon(0) == g["canon"][k][:g["length"][k]].tobytes(), k


def _replay(orc, g, k, check_feat=False):
    lo, hi = int(g["offsets"][k]), int(g["offsets"][k + 1])
    assert orc.reset(0, int(g["seeds"][k]), g["deck0"][k], g["deck1"][k]) == 0
    assert orc.canon_hash(0) == int(g["init_hash"][k])
    for t in range(lo, hi):
        assert np.array_equal(orc.legal_mask(0), g["legal"][t]), (k, t)
        f, r, d = orc.step(0, int(g["action"][t]))
        last_faulted = g["fault"][k] and t == hi - 1
        if last_faulted:
            assert f != 0 or orc.observe(0) is None, (k, t)
            break
        assert f == 0, (k, t, f)
        assert orc.canon_hash(0) == int(g["hash"][t]), (k, t)
        assert orc.obs_hash(0) == int(g["obs"][t]), (k, t)
        assert (r, d) == (int(g["reward"][t]), int(g["done"][t])), (k, t)
        if check_feat:
            assert np.array_equal(orc.features(0).view(np.uint64), g["feat"][t].view(np.uint64)), (k, t)
    return hi - lo


@pytest.mark.parametrize("name,feat", [("trace_random_N12V.npz", False), ("trace_random_N12M.npz", True),
                                       ("trace_random_IRONCLAD.npz", False), ("trace_random_S12.npz", False),
                                       ("trace_pool.npz", False),
                                       ("trace_pool_up.npz", False)])
def test_random_policy_traces(oracle_mod, gold, name, feat):
    g = gold(name)
    orc = oracle_mod.Oracle(1)
    steps = sum(_replay(orc, g, k, feat) for k in range(len(g["seeds"])))
    assert steps == len(g["action"])


def test_expert_bot_traces(oracle_mod, gold):
    """Stormbound.expert_action (games/stormbound.py:563-637) on both sides: the bot's choice (drawn from the
    game stream) and the resulting state at every step."""
    g = gold("trace_expert.npz")
    orc = oracle_mod.Oracle(1)
    n_actions = 0
    for k in range(len(g["seeds"])):
        lo, hi = int(g["offsets"][k]), int(g["offsets"][k + 1])
        assert orc.reset(0, int(g["seeds"][k]), g["deck0"][k], g["deck1"][k]) == 0
        for t in range(lo, hi):
            assert np.array_equal(orc.legal_mask(0), g["legal"][t]), (k, t)
            a, f = orc.expert_action(0)
            if g["action"][t] == 255:   # the reference's bot raised (random.choice([]))
                assert f != 0, (k, t)
                break
            assert f == 0 and a == g["action"][t], (k, t, a, g["action"][t])
            fs, r, d = orc.step(0, a)
            if g["fault"][k] and t == hi - 1:
                assert fs != 0 or orc.observe(0) is None
                break
            assert fs == 0 and orc.canon_hash(0) == int(g["hash"][t]), (k, t)
            assert (r, d) == (int(g["reward"][t]), int(g["done"][t]))
            n_actions += 1
    assert n_actions > 2500


def test_extended_record_pool_traces(oracle_mod, gold):
    """All 109 observable cards (ua20, b005 included) on the extended build -- restored NESTED b005 memories too, whose
    entities live on a deep-copied phantom board in the reference until the next flip (state.h: worlds): every step
    of every game equals the reference, no game is cut short."""
    g = gold("trace_pool_ext.npz")
    orc = oracle_mod.Oracle(1, extended=True)
    steps = sum(_replay(orc, g, k, False) for k in range(len(g["seeds"])))
    assert steps == len(g["action"])


def test_set_iteration_order_known_answers(oracle_mod):
    """cards/s203.py iterates a set of Points: CPython's insertion/probing order, restated in pyset.h."""
    import ctypes
    import json
    import os
    from conftest import GOLD
    L = oracle_mod.lib()
    for idx, exp in json.load(open(os.path.join(GOLD, "set_order_kat.json"))):
        k = np.array(idx, dtype=np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        m = L.orc_pyset_list(_p(k), len(idx), _p(out))
        assert list(out[:m]) == exp


HEURISTIC_FIXTURES = ["trace_heuristic_N12M.npz", "trace_heuristic_S12.npz", "trace_heuristic_IRONCLAD.npz", "trace_heuristic_pool.npz",
                      "trace_heuristic_pool_ext.npz", "trace_heuristic_c5_big.npz"]


def _fixture_build(fixture):
    """Which record a fixture's decks need: 0 standard, 1 extended (ua20 / b005), 2 large (games of the C5 family whose
    nested b005 memories outgrow the extended record -- the product replays such games on libmonsoon_hip_big.so)."""
    return 2 if fixture.endswith("_big.npz") else int(fixture.endswith("_ext.npz"))


@pytest.mark.parametrize("fixture", HEURISTIC_FIXTURES)
def test_heuristic_selfplay_trace(oracle_mod, gold, fixture):
    """Corrected rollout loop vs the reference's HeuristicAgent: every decision's chosen action,
    full score vector (hashed), best score and committed state -- N12M mirror, the Swarm deck S12 (short games,
    s203's set iteration), the reference's default Ironclad-vs-Swarm pair, and per-game random 12-card decks, where
    look-aheads that raise (score 0.0) and committed steps that raise are common.  A game whose committed step raises
    in the reference (hash 0 in the fixture) must fault here at the same decision."""
    g = gold(fixture)
    orc = oracle_mod.Oracle(1, extended=_fixture_build(fixture))
    w = g["w0"]
    faults = g["fault"] if "fault" in g.files else np.zeros(len(g["seeds"]), dtype=np.uint8)
    for k, seed in enumerate(g["seeds"]):
        lo, hi = int(g["offsets"][k]), int(g["offsets"][k + 1])
        if "decks" in g.files:
            deck, deck1 = g["decks"][k]
        else:
            deck = g["deck"]
            deck1 = g["deck1"] if "deck1" in g.files else deck
        orc.reset(0, int(seed), deck, deck1)
        for t in range(lo, hi):
            lf = orc.lookahead_faults(0)
            # no look-ahead hits a limit of this build (255 = illegal; 18 = the recursion guard, which is the reference's
            # RecursionError: the score hash below shows that the reference scored that action 0.0 as well)
            assert not ((lf >= 16) & (lf != 255) & (lf != 18)).any(), (k, t)
            a, scores, _ = orc.decide(0, w)
            legal = ~np.isnan(scores)
            assert a == g["action"][t], (k, t)
            assert int(legal.sum()) == g["nlegal"][t]
            assert oracle_lib.fnv1a64(scores[legal].tobytes()) == int(g["shash"][t]), (k, t)
            assert scores[a] == g["best"][t]
            f = orc.step(0, a)[0]
            if int(g["hash"][t]) == 0:
                assert faults[k] and t == hi - 1 and f != 0, (k, t)
            else:
                assert f == 0 and orc.canon_hash(0) == int(g["hash"][t]), (k, t)
        # and the packaged rollout agrees with the step-by-step one
        orc.reset(0, int(seed), deck, deck1)
        r = orc.rollout(0, w, w, int(g["max_turns"]), trace=True)
        assert r["result"] == g["result"][k] and r["steps"] == hi - lo
        assert np.array_equal(r["actions"], g["action"][lo:hi])
        if not faults[k]:
            assert np.array_equal(r["hashes"], g["hash"][lo:hi])
        else:
            assert r["fault"] != 0 and np.array_equal(r["hashes"][:-1], g["hash"][lo:hi - 1])


def test_heuristic_selfplay_two_weight_vectors(oracle_mod, gold):
    """Different weights for the two agents (reference: agents[adapter.get_current_player()]): the packaged rollout
    picks the mover's vector exactly as the reference does -- actions and committed states of 1 200 decisions."""
    g = gold("trace_heuristic_N12M_2w.npz")
    orc = oracle_mod.Oracle(1)
    assert not np.array_equal(g["w0"], g["w1"])
    for k, seed in enumerate(g["seeds"]):
        lo, hi = int(g["offsets"][k]), int(g["offsets"][k + 1])
        orc.reset(0, int(seed), g["deck"], g["deck1"])
        r = orc.rollout(0, g["w0"], g["w1"], int(g["max_turns"]), trace=True)
        assert r["result"] == g["result"][k] and r["steps"] == hi - lo
        assert np.array_equal(r["actions"], g["action"][lo:hi])
        assert np.array_equal(r["hashes"], g["hash"][lo:hi])


def test_quirk_spell_lands_one_tile_late(oracle_mod):
    """USE action 65+21c+tile executes at tile+1; the last tile is a no-op that
    costs nothing.  Checked as properties of the restatement on a constructed position."""
    from monsoon_amd.cards import deck_indices
    orc = oracle_mod.Oracle(1)
    deck = deck_indices("N12M")
    seen = 0
    for seed in range(200):
        orc.reset(0, seed, deck, deck)
        for _ in range(60):
            la = orc.legal_actions(0)
            uses = [a for a in la if 64 <= a < 148 and (a - 64) % 21 != 0]
            if uses:
                before = orc.canon(0)
                a = uses[-1]
                orc.step(0, a)
                if (a - 64) % 21 == 20:
                    assert orc.canon(0) == before   # fell off the countdown loop: nothing happened
                seen += 1
                break
            orc.step(0, la[0])
            if orc.have_winner(0):
                break
    assert seen > 0


@pytest.mark.parametrize("builds", ["standard+extended", "large"])
def test_reference_unit_tests_as_scenarios(oracle_mod, builds):
    """The reference's OWN tests: 112 `class <ID>Test(CardTestCase)` next to the cards and the
    engine-level BaseTestCase (test.py:53-147: LIFO order of a 16-unit U401 chain reaction, trigger order vs move order,
    respawn), recorded call by call on the reference (oracle/pyref/gen_scenarios.py; all 113 pass there).  For every
    recorded call: load the state the reference had, make the call, land on the reference's canonical state and on the
    reference's order of ability activations."""
    import scenario_lib as S
    from monsoon_amd.cards import CARD_INDEX
    ext_cards = [CARD_INDEX["ua20"], CARD_INDEX["b005"]]
    if builds == "large":   # every scenario on the large record (the replay tier's build)
        big = oracle_mod.Oracle(1, extended=2)
        orcs = {False: big, True: big}
    else:
        orcs = {False: oracle_mod.Oracle(1), True: oracle_mod.Oracle(1, extended=True)}
    n_calls, n_tests, orders = 0, 0, 0
    for case in S.load():
        for k, rec in enumerate(case["records"]):
            ext = S.needs_extended(rec, ext_cards)
            orc = orcs[ext]
            st = rec["before"]
            assert orc.scn_build(0, st["seed"], st["stream_pos"], S.encode_state(st)) == 0, (case["test"], k)
            f, log = orc.scn_op(0, S.encode_op(rec))
            if rec["raised"]:
                assert f != 0, (case["test"], k, rec["op"])
                continue
            assert f == 0, (case["test"], k, rec["op"], f)
            assert orc.canon(0).hex() == rec["after"], (case["test"], k, rec["op"])
            assert log == S.expected_log(rec), (case["test"], k, rec["op"])
            orders += len(log)
            n_calls += 1
        n_tests += 1
    assert n_tests == 113 + 7 and n_calls > 550 and orders > 150   # + the quirk scenarios of SURVEY §0 (G6)


def test_replay_tier_for_games_the_extended_record_cannot_hold():
    """Nested b005 memories are deep copies of the whole game; a record is finite.  The games whose copies outgrow the
    extended record (fault code >= 16) are replayed on the large record (monsoon_amd/fitness.py::replace_capacity_faulted,
    mirrored by tests/oracle_rollout.py): their rows then equal a direct run on the large record, every other row is
    untouched, and on games both records hold the two builds agree."""
    from c5_games import C5_OVERFLOWING, c5_games
    from oracle_rollout import oracle_rollout_fn, oracle_rollout_tier
    idx = C5_OVERFLOWING[:10] + list(range(10))
    m, pairs = c5_games(idx)
    w = np.random.RandomState(2024).uniform(0, 1, 10)[None]   # W0 of tests/test_gpu_parity.py, under which the list was drawn up
    c1, r1, s1, f1 = oracle_rollout_tier(w, m, pairs, 200, 1)
    c2, r2, s2, f2 = oracle_rollout_tier(w, m, pairs, 200, 2)
    ct, rt, st, ft = oracle_rollout_fn(w, m, pairs, 200, want_faults=True)
    over = f1 >= 16
    assert over[:10].sum() >= 9 and over[10:].sum() <= 1
    assert np.array_equal(rt[over], r2[over]) and np.array_equal(st[over], s2[over]) and np.array_equal(ft[over], f2[over])
    assert np.array_equal(rt[~over], r1[~over]) and np.array_equal(st[~over], s1[~over]) and np.array_equal(ft[~over], f1[~over])
    assert np.array_equal(r1[~over], r2[~over]) and np.array_equal(s1[~over], s2[~over])
    assert (ft >= 16).sum() < over.sum() and (ft >= 16).sum() <= 2       # 2063 does not fit 254 slots either
    assert ct[0, 2] == len(idx) and ct[0, 0] == (rt == 0).sum() and ct[0, 1] == (rt == -1).sum()


def test_deck_draw_restatement_vs_numpy(gold):
    """orc_draw_decks (the CPU restatement of monsoon_draw_decks) against numpy's own RandomState(seed).choice(pool, 12,
    replace=False): the committed known answers and 300 seeds drawn by numpy right here."""
    from oracle_rollout import oracle_draw_decks
    from monsoon_amd.cards import draw_random_decks_numpy, observable_pool
    g = gold("deck_draw_kat.npz")
    assert np.array_equal(oracle_draw_decks(g["seeds"], g["pool"]), g["pairs"])
    seeds = np.arange(300, dtype=np.uint32) * np.uint32(40503) + np.uint32(17)
    assert np.array_equal(oracle_draw_decks(seeds, observable_pool()), draw_random_decks_numpy(seeds))


def ulps(a, b):
    """Distance in units of the last place between float64 arrays (same sign assumed where it matters)."""
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    return np.abs(a - b)


def orc_ga(g, tag):
    """The oracle's restatement of Population.generate_offspring on fixture case `tag` -> (w, s, parent, tries, key, pos, has_gauss, gauss)."""
    import ctypes

    class St(ctypes.Structure):
        _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32), ("gauss", ctypes.c_double)]
    L = oracle_lib.lib()
    mu, lam, dim = (int(x) for x in g[f"{tag}_cfg"])
    tau, taup, mins = (float(x) for x in g[f"{tag}_params"])
    st = St()
    key0 = np.ascontiguousarray(g[f"{tag}_st0_key"], dtype=np.uint32)   # kept alive across the memmove
    ctypes.memmove(st.key, key0.ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = int(g[f"{tag}_st0_pos"][0]), int(g[f"{tag}_st0_pos"][1]), float(g[f"{tag}_st0_gauss"][0])
    pw, ps = np.ascontiguousarray(g[f"{tag}_pw"]), np.ascontiguousarray(g[f"{tag}_ps"])
    ow, osg = np.zeros((lam, dim)), np.zeros((lam, dim))
    par, tries = np.zeros(lam, dtype=np.int32), np.zeros(lam, dtype=np.int64)
    L.orc_ga_offspring.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_ga_offspring.restype = None
    p = oracle_lib._p
    L.orc_ga_offspring(ctypes.byref(st), p(pw), p(ps), mu, dim, lam, tau, taup, mins, p(ow), p(osg), p(par), p(tries))
    return ow, osg, par, tries, np.ctypeslib.as_array(st.key).copy(), st.pos, st.has_gauss, st.gauss


def test_ga_offspring_restatement_vs_numpy(gold):
    """the restatement of generate_offspring (randint parent, the copy's discarded uniforms, legacy_gauss with
    its cache, the self-adaptive mutation) against the host GA run by numpy itself (tests/golden/ga_kat.npz): the stream
    state afterwards -- key, position, has_gauss -- is numpy's, i.e. every draw and every accept / reject decision of the
    polar method (1 024 children x 21 normals) was the same; children within 2 ulp (numpy's array exp is its own SIMD
    routine).  And 20 000 raw normal(0, 1) values with the stream position behind each."""
    import ctypes
    g = gold("ga_kat.npz")
    for tag in ("a", "b"):
        ow, osg, par, tries, key, pos, hg, gs = orc_ga(g, tag)
        assert np.array_equal(key, g[f"{tag}_st1_key"]) and [pos, hg] == g[f"{tag}_st1_pos"].tolist()
        assert ulps([gs], g[f"{tag}_st1_gauss"]).max() == 0
        # sigma' = sigma * exp(..): 2 ulp; w' = w + sigma' * z: a few ulp of the STEP, which can be many ulp of a small w'
        assert ulps(osg, g[f"{tag}_ks"]).max() <= 2 and np.abs(ow - g[f"{tag}_kw"]).max() <= 4e-16
        assert par.min() >= 0 and par.max() < int(g[f"{tag}_cfg"][0]) and (np.diff(tries) >= 10).all()
    n = len(g["gauss_values"])
    vals, pos = np.zeros(n), np.zeros(n, dtype=np.int64)
    L = oracle_lib.lib()
    L.orc_np_gauss.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_np_gauss.restype = None
    L.orc_np_gauss(int(g["gauss_seed"][0]), n, oracle_lib._p(vals), oracle_lib._p(pos))
    assert np.array_equal(pos, g["gauss_stream_pos"]) and ulps(vals, g["gauss_values"]).max() == 0
