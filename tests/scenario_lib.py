"""Replayer of the scenario fixtures (tests/golden/scenarios.json.gz): the reference's own 113 unit tests recorded call by
call by oracle/pyref/gen_scenarios.py.  Every record = complete state before, ONE call into the engine, canonical state
after, order in which abilities ran.  encode_state / encode_op turn a record into the int32 streams that
monsoon_amd/csrc/scenario.inc parses (the same code on the CPU oracle and, one lane, on the GPU)."""
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenarios.json.gz")
EXT_CARDS = None   # card indices that need the extended record (filled by needs_extended)


def load():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)


def p_pack(p):   # rules.h p_pack
    return ((p[1] + 1) << 3) | (p[0] + 1)


def _tile(p):
    return -1 if p is None else p[1] * 4 + p[0]


def _entity(d, out):
    out += [d["card"], d["owner"], d["strength"], d.get("movement", 0), 1 if d["ff"] else 0]
    out += d.get("status", [0] * 5)
    out += [_tile(d["position"]), d["damage_taken"], d.get("move_id", 0), 1 if d.get("resolving_play") else 0,
            1 if d["single_use"] else 0]
    path = [p for p in d.get("path", [])]
    out.append(len(path))
    for p in path:
        y = min(max(p[1], -1), 5)   # a path that runs on past a base is clamped exactly as the engine stores it
        out.append(p_pack([p[0], y]))
    mem = d.get("memory", [])
    out.append(len(mem))
    for m in mem:
        _entity(m, out)


def _card(d, out, first_oid):
    """first_oid: the per-player object number (extended record: hand and deck list OBJECTS, which may repeat)."""
    spell = d["kind"] == "spell"
    out += [d["card"], d["cost"], 1 if d["single_use"] else 0, 1 if d["ff"] else 0, -1 if spell else d["strength"],
            0 if spell or d.get("position") is None else 1, d["age"], first_oid]


def encode_state(st):
    out = [st["local_order"], st["cp"], st["phase"], 1 if st["resolving"] else 0, len(st["history"])]
    for owner, card in st["history"]:
        out += [owner, card]
    for p in st["players"]:
        out += [p["base"], p["mana"], p["max_mana"], p["front"], 1 if p["replacable"] else 0, 1 if p["leftmost_movable"] else 0,
                p["faction"], len(p["hand"])]
        local = {}   # object id -> 0, 1, 2 ... in order of first appearance within this player
        for c in p["hand"]:
            _card(c, out, local.setdefault(c["oid"], len(local)))
        out.append(len(p["deck"]))
        for c in p["deck"]:
            _card(c, out, local.setdefault(c["oid"], len(local)))
    for t in st["tiles"]:
        if t is None:
            out.append(0)
        else:
            out.append(1)
            _entity(t, out)
    return np.array(out, dtype=np.int32)


def _recv(on, out):
    if "tile" in on:
        out += [0, on["tile"]]
    elif on["loose"]["kind"] == "spell":
        out += [2, on["loose"]["card"], on["loose"]["owner"]]
    else:
        out.append(1)
        _entity(on["loose"], out)


def _pos(p, out):
    out += [0, 0, 0] if p is None else [1, p[0], p[1]]


def encode_op(rec):
    op, a, out = rec["op"], rec["args"], []
    if op == "Board.spawn_token_unit":
        types = a["types"]
        out += [1, a["owner"], a["position"][0], a["position"][1], a["strength"], -1 if types is None else types[0]]
    elif op == "Board.spawn_token_structure":
        out += [2, a["owner"], a["position"][0], a["position"][1], a["strength"]]
    elif op == "Card.play":
        out.append(3)
        _recv(rec["on"], out)
        _pos(a["position"], out)
    elif op == "Card.activate_ability":
        out.append(4)
        _recv(rec["on"], out)
        _pos(a["position"], out)
        out.append(1 if a["source"] else 0)
    elif op == "Card.deal_damage":
        out.append(5)
        _recv(rec["on"], out)
        out += [a["amount"], 1 if a["pending"] else 0, 1 if a["source"] else 0]
    elif op == "Card.destroy":
        out.append(6)
        _recv(rec["on"], out)
        out.append(1 if a["source"] else 0)
    elif op == "Card.respawn":
        out.append(7)
        _recv(rec["on"], out)
        out += [a["position"][0], a["position"][1], a["strength"]]
    elif op == "Card.command":
        out.append(8)
        _recv(rec["on"], out)
    elif op == "Board.to_next_turn":
        out.append(9)
    elif op == "Player.play":
        out += [10, rec["on"]["player"], a["index"]]
        _pos(a["position"], out)
    elif op == "Player.discard":
        out += [11, rec["on"]["player"], a["index"]]
    else:
        raise ValueError(op)
    return np.array(out, dtype=np.int32)


def needs_extended(rec, ext_cards):
    """True when the record involves ua20 / b005, or lists one card OBJECT twice (hand + deck, or twice in a deck: the
    reference's list.remove() takes the first EQUAL card) -- both need the extended-record build."""
    s = json.dumps(rec["before"]) + json.dumps(rec.get("on", {}))
    if any(f'"card": {c},' in s or f'"card": {c}}}' in s for c in ext_cards):
        return True
    for p in rec["before"]["players"]:
        oids = [c["oid"] for c in p["hand"] + p["deck"]]
        if len(set(oids)) != len(oids):
            return True
    return False


def expected_log(rec):
    return [[c, _tile(p)] for c, p in rec["activations"]]
