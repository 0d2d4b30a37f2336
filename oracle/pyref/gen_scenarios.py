#!/usr/bin/env python3
"""The reference's OWN tests as data: one fixture record per engine call its test bodies make.

TEST INFRASTRUCTURE (build container only; needs /root/reference).  Writes tests/golden/scenarios.json.gz.

The reference's author wrote 113 unittest cases: `class <ID>Test(CardTestCase)` next to every card
(e.g. cards/u007.py:24-68) and the engine-level `BaseTestCase.test_ability` (test.py:53-147: LIFO trigger order of a
16-unit chain, trigger-vs-move order, respawn).  Their bodies are arbitrary Python (attribute pokes, list surgery,
monkey-patched hooks), so they are not replayed line by line.  Instead every test is RUN here, on the reference, with
its engine classes instrumented: each call the test body makes directly into the engine (card.play, activate_ability,
deal_damage, destroy, command, respawn, spawn_token_*, to_next_turn, Player.play/discard ...) is recorded as

    state before (complete: board entities with their hidden attributes, players, hands, decks, stream position)
    the call (method + arguments as plain data)
    canonical state after (harness.canon, the same byte record every parity test uses)
    the (card, position) sequence of activate_ability calls that ran inside it (trigger resolution order)

Whatever the test body does between two such calls is absorbed by the next record's "state before".  A replayer
(tests/scenario_lib.py) loads each state, performs the one call and must land on the recorded state and order -- on
the CPU oracle and, through monsoon_state_load + monsoon_debug_op, on the GPU.  Every reference test is also required
to PASS here, so matching the recorded states means satisfying the author's assertions.

The stream: CardTestCase seeds RandomState(int(time.time())) (test.py:15); time.time is pinned, and after the fixture's
__init__ the shared RandomState object is re-seeded in place with a per-test seed, so a record stores (seed, absolute
position) instead of 624 words.
"""
import functools
import importlib
import io
import json
import os
import sys
import contextlib
import time as _time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import harness as H  # noqa: E402  (sets up the reference import environment)

import cards as refcards  # noqa: E402
import test as reftest  # noqa: E402  (the reference's test.py)
from board import Board  # noqa: E402
from card import Card  # noqa: E402
from enums import StatusEffect  # noqa: E402
from player import Player  # noqa: E402
from point import Point  # noqa: E402
from spell import Spell  # noqa: E402
from structure import Structure  # noqa: E402
from unit import Unit  # noqa: E402

OUT = os.path.join(H.REPO, "tests", "golden", "scenarios.json.gz")
PINNED_TIME = 1_700_000_000
STATUS = (StatusEffect.FROZEN, StatusEffect.POISONED, StatusEffect.CONFUSED, StatusEffect.DISABLED, StatusEffect.VITALIZED)

# weight -> age (the records store Card.weight as an integer number of reweights, state.h)
_WT = [1.0]
for _ in range(255):
    _WT.append(_WT[-1] * 1.6 + 100)
_AGE = {w: k for k, w in enumerate(_WT)}


class Unsupported(Exception):
    pass


def pt(p):
    return None if p is None else [int(p.x), int(p.y)]


def card_index(c):
    if type(c) is Unit and len(c.unit_types) != 1:
        raise Unsupported("token unit with several unit types (only the tests build those)")
    return H.card_index(c)


def card_desc(c):
    """A card object that is not on the board (hand, deck, or about to be played)."""
    d = {"card": card_index(c), "cost": int(c.cost), "single_use": bool(c.is_single_use),
         "ff": bool(getattr(c, "fixedly_forward", False)), "kind": "spell" if isinstance(c, Spell) else ("unit" if isinstance(c, Unit) else "structure")}
    if not isinstance(c, Spell):
        d["strength"] = int(c.strength)
        d["position"] = pt(c.position)
    if isinstance(c, Unit):
        d["movement"] = int(c.movement)
    w = float(c.weight)
    if w == 0.0:
        d["age"] = 0     # Card.__init__: weight 0 until a Player adopts it; never read before it is set
    elif w in _AGE:
        d["age"] = _AGE[w]
    else:
        raise Unsupported(f"weight {w} is not f^k(1)")
    return d


def ent_desc(e, depth=0):
    """An entity object with every attribute the engine reads."""
    d = card_desc(e)
    d["owner"] = int(e.player.order)
    d["damage_taken"] = int(e.damage_taken)
    if isinstance(e, Unit):
        d["status"] = [e.status_effects.count(s) for s in STATUS]
        d["path"] = [pt(p) for p in e.path]
        d["move_id"] = int(e.move_id) & 0xFF
        d["resolving_play"] = bool(e.resolving_play)
        d["token_types"] = [int(t) for t in e.unit_types] if type(e) is Unit else None
    mem = getattr(e, "ability_remembered", None)
    if mem:
        if depth > 0:
            raise Unsupported("nested b005 memory in a scenario")
        d["memory"] = [ent_desc(m, depth + 1) for m in mem]
    return d


def stream_position(rs, seed):
    """Absolute index of the next u32 of RandomState(seed) that `rs` (the same stream, later) will hand out."""
    _, key, pos = rs.get_state()[:3]
    probe = np.random.RandomState(seed)
    for k in range(0, 64):
        pkey = probe.get_state()[1]
        if np.array_equal(pkey, key):
            # a fresh RandomState holds the seeded (untwisted) key at pos 624; block b's outputs are handed out while
            # the key is the (b+1)-times-twisted one
            return (k - 1) * 624 + int(pos) if k > 0 else 0
        probe.randint(0, 2**32, size=624, dtype=np.uint32)   # consume exactly one block -> one twist
        # after consuming 624 from the initial state (pos 624) the generator twisted once and sits at pos 624 again
    raise Unsupported("stream position not found")


def state_desc(board, seed):
    players = {int(board.local.order): board.local, int(board.remote.order): board.remote}
    oids = {}   # object identity of hand / deck cards: list.remove() takes the first EQUAL card (unit.py:25-26), so with
    #             duplicate ids in a deck the same object can sit in the hand and the deck, or twice in a deck
    st = {"local_order": int(board.local.order), "cp": int(board.current_player.order), "phase": int(board.phase.value) if hasattr(board.phase, "value") else int(board.phase),
          "resolving": bool(board.is_resolving_trigger), "triggers": len(board.triggers),
          "history": [[int(c.player.order), card_index(c)] for c in board.history[-4:]],
          "players": [], "tiles": [], "seed": seed, "stream_pos": stream_position(board.random, seed)}
    if board.triggers:
        raise Unsupported("pending triggers between engine calls")
    for order in (0, 1):
        p = players[order]
        st["players"].append({"base": int(p.strength), "mana": int(p.current_mana), "max_mana": int(p.max_mana), "front": int(p.front_line),
                              "replacable": bool(p.replacable), "leftmost_movable": bool(p.leftmost_movable), "faction": int(p.faction),
                              "hand": [dict(card_desc(c), oid=oids.setdefault(id(c), len(oids))) for c in p.hand],
                              "deck": [dict(card_desc(c), oid=oids.setdefault(id(c), len(oids))) for c in p.deck]})
    for y in range(5):
        for x in range(4):
            e = board.board[y][x]
            st["tiles"].append(None if e is None else ent_desc(e))
    return st


class FakeGame:
    """What harness.canon needs of a Stormbound object."""

    def __init__(self, board):
        self.board, self.player, self.random = board, 1 if int(board.local.order) == 0 else -1, board.random


class Tracer:
    def __init__(self):
        self.depth = 0
        self.records = []
        self.board = None
        self.seed = None
        self.activations = []
        self.skipped = []

    def where(self, obj):
        """How the replayer finds `obj`: an entity on the board by its tile, else a loose card by description."""
        b = self.board
        if isinstance(obj, (Unit, Structure)):
            for y in range(5):
                for x in range(4):
                    if b.board[y][x] is obj:
                        return {"tile": y * 4 + x}
            d = ent_desc(obj) if obj.player is not None else None
            if d is None:
                raise Unsupported("loose entity without a player")
            return {"loose": d}
        if isinstance(obj, Spell):
            d = card_desc(obj)
            d["owner"] = int(obj.player.order)
            return {"loose": d}
        raise Unsupported(f"receiver {type(obj)}")

    def wrap(self, cls, name, argspec):
        orig = cls.__dict__[name]
        tr = self

        @functools.wraps(orig)
        def wrapper(obj, *args, **kwargs):
            top = tr.depth == 0 and tr.board is not None
            rec = None
            if top:
                try:
                    rec = {"op": f"{'Board' if isinstance(obj, Board) else 'Player' if isinstance(obj, Player) else 'Card'}.{name}",
                           "args": argspec(tr, obj, args, kwargs), "before": state_desc(tr.board, tr.seed)}
                    if not isinstance(obj, Board):
                        rec["on"] = {"player": int(obj.order)} if isinstance(obj, Player) else tr.where(obj)
                except Unsupported as e:
                    tr.skipped.append(f"{name}: {e}")
                    rec = None
                tr.activations = []
            tr.depth += 1
            try:
                return orig(obj, *args, **kwargs)
            finally:
                tr.depth -= 1
                if top and rec is not None:
                    try:
                        for pl in (tr.board.local, tr.board.remote):
                            if any(float(c.weight) not in _AGE for c in pl.deck):
                                # a card object the test built by hand (Card.__init__: weight 0) reached a deck without ever
                                # being drawn: 0 * 1.6 + 100 ... is outside the weights a game can produce
                                raise Unsupported("a deck card's weight is not f^k(1) (hand-built card, weight 0)")
                        rec["after"] = H.canon(FakeGame(tr.board)).hex()
                        rec["activations"] = tr.activations
                        rec["raised"] = sys.exc_info()[0] is not None
                        tr.records.append(rec)
                    except AssertionError:
                        tr.skipped.append(f"{name}: state after holds a token with several unit types")
                    except Unsupported as e:
                        tr.skipped.append(f"{name}: {e}")
        setattr(cls, name, wrapper)

    def wrap_activation(self, cls):
        """activate_ability of a card class (already inside the reference's trigger-drain wrapper): log the order."""
        orig = cls.__dict__["activate_ability"]
        tr = self

        @functools.wraps(orig)
        def wrapper(obj, *args, **kwargs):
            if tr.board is not None:
                tr.activations.append([H.CARD_INDEX[obj.card_id], pt(getattr(obj, "position", None))])
            return orig(obj, *args, **kwargs)
        cls.activate_ability = wrapper


def a_none(tr, obj, args, kw):
    return {}


def a_point(tr, obj, args, kw):
    p = args[0] if args else kw.get("position")
    return {"position": pt(p)}


def a_ability(tr, obj, args, kw):
    p = args[0] if args else kw.get("position")
    src = args[1] if len(args) > 1 else kw.get("source")
    return {"position": pt(p), "source": src is not None}


def a_damage(tr, obj, args, kw):
    amount = args[0] if args else kw["amount"]
    pending = args[1] if len(args) > 1 else kw.get("pending_destroy", False)
    src = args[2] if len(args) > 2 else kw.get("source")
    return {"amount": int(amount), "pending": bool(pending), "source": src is not None}


def a_source(tr, obj, args, kw):
    src = args[0] if args else kw.get("source")
    return {"source": src is not None}


def a_respawn(tr, obj, args, kw):
    return {"position": pt(args[0]), "strength": int(args[1])}


def a_spawn(tr, obj, args, kw):
    player, position, strength = args[0], args[1], args[2]
    types = args[3] if len(args) > 3 else kw.get("types")
    return {"owner": int(player.order), "position": pt(position), "strength": int(strength),
            "types": None if types is None else [int(t) for t in types]}


def a_pplay(tr, obj, args, kw):
    return {"index": int(args[0]), "position": pt(args[1] if len(args) > 1 else kw.get("position"))}


def a_discard(tr, obj, args, kw):
    target = args[0]
    idx = [i for i, c in enumerate(obj.hand) if c is target]
    if not idx:
        raise Unsupported("discard of a card that is not in the hand")
    return {"index": idx[0]}


def install(tr):
    tr.wrap(Board, "spawn_token_unit", a_spawn)
    tr.wrap(Board, "spawn_token_structure", a_spawn)
    tr.wrap(Board, "to_next_turn", a_none)
    tr.wrap(Player, "play", a_pplay)
    tr.wrap(Player, "discard", a_discard)
    for cls, names in ((Unit, {"play": a_point, "deal_damage": a_damage, "destroy": a_source, "command": a_none, "respawn": a_respawn}),
                       (Structure, {"play": a_point, "deal_damage": a_damage, "destroy": a_source, "respawn": a_respawn}),
                       (Spell, {"play": a_point})):
        for n, spec in names.items():
            tr.wrap(cls, n, spec)
    seen = set()
    for name in dir(refcards):
        cls = getattr(refcards, name)
        if isinstance(cls, type) and issubclass(cls, Card) and "activate_ability" in cls.__dict__ and cls not in seen:
            seen.add(cls)
            tr.wrap_activation(cls)
            tr.wrap(cls, "activate_ability", a_ability)   # a direct call from a test body is an engine call too


def run_case(tr, case_cls, seed, body=None):
    _time.time = lambda: PINNED_TIME
    t = case_cls("test_ability")
    t.board.random.seed(seed)   # the ONE RandomState object shared by board, players and cards
    tr.board, tr.seed, tr.records, tr.skipped = t.board, seed, [], []
    with contextlib.redirect_stdout(io.StringIO()):
        if body is None:
            t.test_ability()    # raises if one of the author's assertions fails
        else:
            body(t)
    tr.board = None
    return tr.records, tr.skipped


# ---- G6: the quirks of SURVEY.md §0 as scenarios of the same kind (written here, run on the reference) -----------------
def _card(cid, player):
    c = getattr(refcards, cid.upper())()
    c.player = player
    return c


def _raises(fn):
    try:
        fn()
    except Exception:  # noqa: BLE001
        return True
    return False


def quirk_zombie_path(t):
    """fact #5 (unit.py:148,223): destroy() rebinds self.path but move() keeps iterating the old list.  U206 (2/5/2,
    ON_DEATH: own base -3) played at (0,1) under a 10-strength token: it dies on the first step, attacks again with
    strength 0 on the second (the token's cached strength still hurts nothing) and is destroyed a second time --
    two ON_DEATH firings.  SURVEY §6: own base 14, token 5, enemy base 20."""
    t.board.spawn_token_unit(t.remote, Point(0, 0), 10)
    _card("u206", t.local).play(Point(0, 1))
    assert (t.local.strength, t.board.at(Point(0, 0)).strength, t.remote.strength) == (14, 5, 20)


def quirk_u310_raises(t):
    """cards/u310.py:31-40: `target` is unbound when no bordering enemy can be pushed -> UnboundLocalError."""
    t.board.spawn_token_unit(t.remote, Point(1, 3), 4)   # not bordering (2,1)
    assert _raises(lambda: _card("u310", t.local).play(Point(2, 1)))


def quirk_s101_raises(t):
    """cards/s101.py:22: sorted(...)[0] on an empty list when the caster has no unit -> IndexError (after the mana gain)."""
    assert _raises(lambda: _card("s101", t.local).play(None))
    assert t.local.current_mana == t.local.max_mana + 13


def quirk_u017_raises(t):
    """cards/u017.py:32: random.choice([]) when a chosen spell needs a target and none exists."""
    t.local.hand = t.local.hand[:3] + [_card("s001", t.local)]   # s001 needs an enemy unit; the board has none
    assert _raises(lambda: _card("u017", t.local).play(Point(1, 4)))


def quirk_opponent_is_self_after_flip(t):
    """fact #3 (player.py:42-44): Player.opponent is board.remote for FIRST and board.local for SECOND, so after one
    flip the SECOND player's opponent is itself.  UE42 (AFTER_SURVIVING: opponent base -2, self +2) of the second
    player, damaged while the second player is board.local: its OWN base pays."""
    t.board.flip()
    second = t.board.local
    assert int(second.order) == 1 and second.opponent is second
    ue42 = _card("ue42", second)
    ue42.play(Point(1, 4))
    base_before = second.strength
    t.board.at(ue42.position).deal_damage(3)
    assert second.strength == base_before - 2 and t.board.remote.strength == 20


def quirk_snapshot_iteration(t):
    """fact #6 (board.py:141-143): turn-start movement iterates a SNAPSHOT of unit objects.  The mover's UA04 (BEFORE_MOVING:
    the weakest unit of the side with more units dies) kills the mover's own UA07 earlier in the same loop; the dead UA07
    is still processed: set_path() gives it a path again, its BEFORE_MOVING ability draws from the stream, and since
    destroy() leaves its strength alone it walks back onto the board."""
    t.board.spawn_token_unit(t.local, Point(2, 2), 9)          # an enemy of the mover, so that sides differ in size
    ua04 = _card("ua04", t.remote)
    ua07 = _card("ua07", t.remote)
    t.board.set(Point(3, 3), ua04)
    t.board.set(Point(0, 1), ua07)
    t.board.to_next_turn()                                      # the remote player becomes the mover; its scan runs y 4->0, x 3->0
    assert t.board.at(Point(0, 2)) is ua07 and ua07.strength == 5   # killed by ua04, then moved by the snapshot loop


def quirk_status_multiset(t):
    """fact #4 (unit.py:239-268): statuses are lists, removal pops ONE instance.  Frozen twice = two skipped turns."""
    u = t.board.spawn_token_unit(t.remote, Point(1, 1), 6)
    u.freeze()
    u.freeze()
    for expect_y in (1, 1, 2):
        t.board.to_next_turn()   # remote moves
        assert u.position.y == expect_y, (u.position, expect_y)
        t.board.to_next_turn()   # local moves (nothing of its own on the board)


QUIRKS = [quirk_zombie_path, quirk_u310_raises, quirk_s101_raises, quirk_u017_raises, quirk_opponent_is_self_after_flip,
          quirk_snapshot_iteration, quirk_status_multiset]


def main():
    tr = Tracer()
    install(tr)
    cases = [("base", reftest.BaseTestCase)]
    for cid in H.CARD_IDS:
        mod = importlib.import_module(f"cards.{cid}")
        cases.append((cid, getattr(mod, cid.upper() + "Test")))
    cases += [(q.__name__, q) for q in QUIRKS]
    out, n_ops, ops = [], 0, {}
    for k, (name, cls) in enumerate(cases):
        if name.startswith("quirk_"):
            recs, skipped = run_case(tr, reftest.CardTestCase, 7000 + k, body=cls)
        else:
            recs, skipped = run_case(tr, cls, 7000 + k)
        for r in recs:
            ops[r["op"]] = ops.get(r["op"], 0) + 1
        n_ops += len(recs)
        out.append({"test": name, "seed": 7000 + k, "records": recs, "skipped": skipped})
        if skipped:
            print(name, "skipped:", skipped)
    print(f"{len(cases) - len(QUIRKS)} reference tests passed, {len(QUIRKS)} quirk scenarios hold; {n_ops} engine calls recorded")
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]):
        print(f"  {v:4d} {k}")
    import gzip
    with gzip.GzipFile(OUT, "wb", mtime=0) as f:   # deterministic bytes
        f.write(json.dumps(out, separators=(",", ":")).encode())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
