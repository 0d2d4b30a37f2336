"""Card ids <-> table indices, and the fixed decks of the benchmark configurations.

The table (monsoon_amd/card_ids.json, monsoon_amd/csrc/card_table.inc) is generated from the
reference's card constructors by oracle/pyref/gen_card_table.py; index = position in the sorted
id list (so index order == card_id string order, which the observation's deck sort relies on,
games/stormbound.py:441).
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "card_ids.json")) as _f:
    CARD_META = json.load(_f)
CARD_IDS = [c["id"] for c in CARD_META]
CARD_INDEX = {cid: i for i, cid in enumerate(CARD_IDS)}

# Cards that need the extended per-game record (BatchEngine(extended=True), libmonsoon_hip_ext.so):
# ua20 grows the deck, b005 keeps deep copies of its neighbours.  The standard build refuses them.
UNSUPPORTED = frozenset({"ua20", "b005"})
NEEDS_EXTENDED = UNSUPPORTED
# int(card) raises for these (card.py:46): every observation containing one faults.
FAULT_CARDS = frozenset({"up01", "up02", "up03"})

# SURVEY.md Appendix G / games/stormbound.py:295-302
DECKS = {
    "N12V": "u001 u002 u003 u019 u020 u025 u027 u030 u031 u032 u069 b001".split(),
    "N12M": "u001 u007 u020 u021 u026 u053 u061 ua07 ue01 s001 s012 b002".split(),
    "S12": "u040 u206 u211 u212 u216 u217 ue21 ue22 ut21 ut22 s203 b203".split(),
    "IRONCLAD": "ua07 u007 u306 u061 b304 u305 u320 u302 u313 ua02 ut32 u316".split(),
    "SWARM": "ua07 u007 u001 u053 ue01 u211 u206 u071 u020 s013 b001 u061".split(),
}


def deck_indices(deck):
    """12 card ids (or a DECKS key) -> uint8[12] table indices."""
    if isinstance(deck, str):
        deck = DECKS[deck]
    arr = np.array([CARD_INDEX[c] if isinstance(c, str) else int(c) for c in deck], dtype=np.uint8)
    if arr.shape != (12,):
        raise ValueError("a deck has exactly 12 cards")
    return arr


def supported_pool(include_fault_cards=False, extended=False):
    return [c for c in CARD_IDS if (extended or c not in UNSUPPORTED) and (include_fault_cards or c not in FAULT_CARDS)]


def needs_extended(decks):
    """True if any deck (ids or indices) holds a card that only the extended build supports."""
    idx = {CARD_INDEX[c] for c in NEEDS_EXTENDED}
    return any((CARD_INDEX[c] if isinstance(c, str) else int(c)) in idx for c in np.asarray(decks, dtype=object).ravel())


def needs_extended_each(deck_pairs):
    """bool[n]: which deck pairs [n][2][12] (card indices) hold a card that only the extended build supports."""
    pairs = np.asarray(deck_pairs, dtype=np.uint8).reshape(-1, 24)
    idx = np.array(sorted(CARD_INDEX[c] for c in NEEDS_EXTENDED), dtype=np.uint8)
    return np.isin(pairs, idx).any(axis=1)


# Configuration C5: every game draws its two decks from the 109 observable cards -- every card but
# up01/up02/up03, whose int(card) raises (card.py:46) and would end half of the games before their first step -- with a
# pre-stream of its own: RandomState(seed ^ C5_STREAM_XOR).choice(pool, 12, replace=False), P1's deck first.
RANDOM_DECK = "random109"
C5_STREAM_XOR = 0x9E3779B9


def observable_pool():
    return np.array([i for i, c in enumerate(CARD_IDS) if c not in FAULT_CARDS], dtype=np.uint8)


def draw_random_decks_numpy(seeds, pool=None):
    """The specification of monsoon_draw_decks, by numpy itself (150 us per game: tests and small schedules only)."""
    pool = observable_pool() if pool is None else np.asarray(pool, dtype=np.uint8)
    out = np.zeros((len(seeds), 2, 12), dtype=np.uint8)
    for k, s in enumerate(seeds):
        rs = np.random.RandomState(int(s))
        out[k, 0], out[k, 1] = rs.choice(pool, 12, replace=False), rs.choice(pool, 12, replace=False)
    return out
