"""Deck schedule of the GA: generate_random_deck and the three-phase DeckEvolutionConfig.

Mirrors /root/reference/utils.py:26-242 over card ids instead of card objects.  The reference draws from Python's
global `random` module (unseeded); here every draw goes through an injectable `random.Random`, so that a schedule is
reproducible.  With the same generator state the selections are the reference's own (same `sample` / `choices` /
`random` calls over the same lists in the same order), pinned by tests/golden/deck_schedule.json.
"""
import random as _random

from .cards import CARD_IDS, CARD_META

NEUTRAL, WINTER, SWARM, IRONCLAD, SHADOWFEN = range(5)   # enums.py:44-49
FACTION_OF = {c["id"]: c["faction"] for c in CARD_META}


def available_cards(faction):
    """utils.py:63-88: every card class of the faction or NEUTRAL, in dir(cards) order (= sorted ids)."""
    return [c for c in CARD_IDS if FACTION_OF[c] in (faction, NEUTRAL)]


def generate_random_deck(faction, original=None, preserve_ratio=0.0, rng=_random):
    """utils.py:26-119.  `original`: list of card ids; returns 12 card ids (fewer only if the pool is empty)."""
    preserve_ratio = max(0.0, min(1.0, preserve_ratio))
    if original and preserve_ratio > 0.0:
        cards_to_preserve = min(int(12 * preserve_ratio), len(original), 12)
        if preserve_ratio == 1.0:
            preserved = list(original[:12])
            if len(preserved) < 12:
                cards_needed = 12 - len(preserved)
            else:
                return preserved
        else:
            preserved = rng.sample(list(original), cards_to_preserve)
            cards_needed = 12 - cards_to_preserve
    else:
        preserved = []
        cards_needed = 12
    if cards_needed > 0:
        pool = available_cards(faction)
        if not pool:
            return preserved
        if cards_needed > len(pool):
            selected = rng.choices(pool, k=cards_needed)   # not enough cards: duplicates allowed
        else:
            selected = rng.sample(pool, cards_needed)
        return preserved + selected
    return preserved


class DeckEvolutionConfig:
    """utils.py:121-242: exploit (archetypes) -> explore (growing share of random cards) -> balance (steady mix)."""

    def __init__(self, player1_archetype, player2_archetype, exploit_generations=30, explore_generations=30,
                 max_random_ratio=0.5, balance_archetype_ratio=0.7, seed=None):
        self.player1_archetype = list(player1_archetype)
        self.player2_archetype = list(player2_archetype)
        self.exploit_generations = exploit_generations
        self.explore_generations = explore_generations
        self.max_random_ratio = max_random_ratio
        self.balance_archetype_ratio = balance_archetype_ratio
        self.player1_faction = FACTION_OF[self.player1_archetype[0]] if self.player1_archetype else NEUTRAL
        self.player2_faction = FACTION_OF[self.player2_archetype[0]] if self.player2_archetype else NEUTRAL
        self.rng = _random.Random(seed) if seed is not None else _random

    def get_deck_configuration(self, generation):
        if generation < self.exploit_generations:
            return list(self.player1_archetype), list(self.player2_archetype)
        if generation < self.exploit_generations + self.explore_generations:
            progress = (generation - self.exploit_generations) / self.explore_generations
            ratio = progress * self.max_random_ratio
            d1 = generate_random_deck(self.player1_faction, self.player1_archetype, 1.0 - ratio, self.rng)
            d2 = generate_random_deck(self.player2_faction, self.player2_archetype, 1.0 - ratio, self.rng)
            return d1, d2
        use1 = self.rng.random() < self.balance_archetype_ratio
        use2 = self.rng.random() < self.balance_archetype_ratio
        d1 = list(self.player1_archetype) if use1 else generate_random_deck(self.player1_faction, rng=self.rng)
        d2 = list(self.player2_archetype) if use2 else generate_random_deck(self.player2_faction, rng=self.rng)
        return d1, d2

    def is_static(self, generation):
        """True while every game of the generation gets the same pair (exploit phase)."""
        return generation < self.exploit_generations

    def get_phase_info(self, generation):
        if generation < self.exploit_generations:
            phase, ratio = "Exploit", 0.0
        elif generation < self.exploit_generations + self.explore_generations:
            phase = "Explore"
            ratio = (generation - self.exploit_generations) / self.explore_generations * self.max_random_ratio
        else:
            phase, ratio = "Balance", 1.0 - self.balance_archetype_ratio
        return {"phase": phase, "generation": generation, "random_ratio": ratio,
                "exploit_complete": generation >= self.exploit_generations,
                "explore_complete": generation >= self.exploit_generations + self.explore_generations}
