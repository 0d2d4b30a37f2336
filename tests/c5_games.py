"""The C5 game family shared by the GPU parity tests, the oracle tests and scripts/c5_capacity.py."""
import numpy as np


def c5_games(indices):
    """Game k of the C5 family used by these tests and scripts/c5_capacity.py: seed 90000 + k, two 12-card decks drawn from
    the 109 observable cards by RandomState(k ^ 0x9E3779B9)."""
    from monsoon_amd.cards import CARD_IDS
    pool = np.array([i for i, c in enumerate(CARD_IDS) if c not in ("up01", "up02", "up03")], dtype=np.uint8)
    pairs = np.zeros((len(indices), 2, 12), dtype=np.uint8)
    m = np.zeros(len(indices), dtype=[("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])
    for j, k in enumerate(indices):
        rs = np.random.RandomState(int(k) ^ 0x9E3779B9)
        pairs[j, 0], pairs[j, 1] = rs.choice(pool, 12, replace=False), rs.choice(pool, 12, replace=False)
        m["seed"][j], m["deck"][j] = 90000 + int(k), j
    return m, pairs


# C5 games (of the first 32 768, weight vector W0) whose nested b005 memories did not fit round 2's extended record of 128 entity
# slots / 16 memory lists (scripts/c5_capacity.py); 2063 and 6149 do not fit the large record either
C5_OVERFLOWING = [264, 1374, 2063, 2103, 2458, 2649, 3525, 3540, 3691, 4215, 4593, 5070, 6011, 6149, 6592, 7552, 8558, 9657, 10155, 10993]
