"""monsoon_amd: MI355X-native batched Stormbound engine behind the reference's game / fitness API.

Product code only.  Nothing here imports oracle/ (the CPU replay oracle is test infrastructure).
"""
from ._lib import MonsoonError  # noqa: F401
from .cards import CARD_IDS, CARD_INDEX, DECKS, deck_indices  # noqa: F401

__all__ = ["MonsoonError", "CARD_IDS", "CARD_INDEX", "DECKS", "deck_indices"]
