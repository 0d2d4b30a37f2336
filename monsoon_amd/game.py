"""Game -- the reference's AbstractGame surface (games/abstract_game.py:4-101, implemented by
games/stormbound.py:121-250) as a batch=1 view over the HIP engine.

    g = Game(seed)                      # default decks = games/stormbound.py:295-302
    obs, reward, done = g.step(action)  # obs (27,5,4) int32, reward*10, done
    g.to_play(); g.legal_actions(); g.reset(); g.close()

`g.env` exposes the attributes evo/game_adapter.py reaches into: get_observation(), have_winner(),
player, to_play(), legal_actions().  A step the reference would raise on raises StepFault here.
"""
import numpy as np

from .cards import CARD_IDS, deck_indices
from .engine import BatchEngine

FACTION = {"NEUTRAL": 0, "WINTER": 1, "SWARM": 2, "IRONCLAD": 3, "SHADOWFEN": 4}   # enums.py:44-49


class StepFault(RuntimeError):
    """The reference raises a Python exception on this transition (fault code in .code)."""

    def __init__(self, code, action):
        super().__init__(f"engine fault {code} on action {action}")
        self.code = code


def action_to_string(a):
    """Same wording as the reference's actions.txt / enums.py:10-36."""
    if a < 64:
        return f"Place unit or structure card at index {a // 16} of hand at ({(a % 16) % 4}, {4 - (a % 16) // 4})"
    if a < 148:
        c, i = divmod(a - 64, 21)
        if i == 0:
            return f"Use spell card at index {c} of hand with no target"
        return f"Use spell card at index {c} of hand at ({(i - 1) % 4}, {4 - (i - 1) // 4})"
    if a < 152:
        return f"Replace card at index {a - 148} of hand"
    if a < 155:
        return f"Move card at index {a - 151} of hand to leftmost"
    return "Pass the turn"


class Stormbound:
    """games/stormbound.py:292-373 over one device-resident game."""

    def __init__(self, seed, deck0="IRONCLAD", deck1="SWARM", faction0=3, faction1=2, device=0, engine=None):
        if seed is None:
            seed = int(np.random.randint(0, 2**32, dtype=np.uint64))   # RandomState(None): OS entropy in the reference
        self.seed = int(seed) & 0xFFFFFFFF
        self._decks = np.stack([deck_indices(deck0), deck_indices(deck1)])
        self._factions = np.array([[faction0, faction1]], dtype=np.uint8)
        self._eng = engine or BatchEngine(1, device=device)
        self._eng.reset(np.array([self.seed], dtype=np.uint32), self._decks[None], self._factions)

    @property
    def player(self):
        return 1 if self.to_play() == 0 else -1

    def to_play(self):
        return int(self._eng.status()[0, 0])

    def reset(self):
        return self.get_observation()

    def have_winner(self):
        return bool(self._eng.status()[0, 1])

    def legal_actions(self):
        return self._eng.legal_actions(0)

    def get_observation(self):
        obs, raises = self._eng.observe()
        if raises[0]:
            raise ValueError("invalid literal for int() with base 16 (card.py:46: up01/up02/up03 visible)")
        return obs[0]

    def step(self, action):
        reward, done, fault = self._eng.step(np.array([action], dtype=np.uint8))
        if fault[0]:
            raise StepFault(int(fault[0]), action)
        return self.get_observation(), int(reward[0]), bool(done[0])

    def expert_action(self):
        """games/stormbound.py:563-637 (draws from the game's stream)."""
        action, fault = self._eng.expert_action()
        if fault[0]:
            raise StepFault(int(fault[0]), "expert_action")
        return int(action[0])

    def state_record(self):
        return self._eng.export(0)

    def clone(self):
        """copy.deepcopy(game) (evo/game_adapter.py:280-287 clone_state): a second device-resident game holding the complete
        state -- stream position included, so the clone continues exactly like the original would."""
        other = Stormbound.__new__(Stormbound)
        other.seed, other._decks, other._factions = self.seed, self._decks, self._factions
        other._eng = BatchEngine(1, device=self._eng.device, extended=self._eng.extended)
        other._eng.load_state(0, self._eng.save_state(0))
        return other

    def features(self):
        """StateFeatures(observation, to_play).get_feature_vector() (evo/features.py:327-342) of the current state."""
        return self._eng.features()[0]

    def close(self):
        self._eng.close()

    def render(self):
        obs = self.get_observation()
        rows = []
        for y in range(5):
            cells = []
            for x in range(4):
                if obs[1, y, x] != -1:
                    cells.append(f"L{obs[1, y, x]:3d}")
                elif obs[17, y, x] != -1:
                    cells.append(f"R{obs[17, y, x]:3d}")
                elif obs[5, y, x] != -1:
                    cells.append(f"l{obs[5, y, x]:3d}")
                elif obs[21, y, x] != -1:
                    cells.append(f"r{obs[21, y, x]:3d}")
                else:
                    cells.append("  . ")
            rows.append(" ".join(cells))
        print(f"remote base {obs[23, 0, 0]}\n" + "\n".join(rows) + f"\nlocal base {obs[14, 0, 0]}  mana {obs[13, 0, 0]}")


class Game:
    def __init__(self, seed=None, **kw):
        self.env = Stormbound(seed, **kw)

    def step(self, action):
        observation, reward, done = self.env.step(action)
        return observation, reward * 10, done       # games/stormbound.py:139-140

    def to_play(self):
        return self.env.to_play()

    def legal_actions(self):
        return self.env.legal_actions()

    def reset(self):
        return self.env.reset()

    def render(self):
        self.env.render()

    def close(self):
        self.env.close()

    def clone(self):
        """StormboundAdapter.clone_state (evo/game_adapter.py:280-287): an independent copy of the whole game."""
        other = Game.__new__(Game)
        other.env = self.env.clone()
        return other

    def expert_agent(self):
        return self.env.expert_action()             # games/stormbound.py:227-235

    def action_to_string(self, action_number):
        return action_to_string(action_number)


def card_name(index):
    return CARD_IDS[index] if index < len(CARD_IDS) else ("token-structure" if index == 128 else f"token-unit-{index - 112}")
