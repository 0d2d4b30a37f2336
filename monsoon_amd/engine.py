"""BatchEngine: numpy-facing wrapper of the C ABI (one handle = one GPU, a batch of games)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import Config, Match, MonsoonError, Stats


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class BatchEngine:
    def __init__(self, max_games, device=0, lanes_per_game=0, stack_bytes=0, extended=False):
        """extended=True loads the build with the larger per-game record (decks holding ua20 / b005), extended=2 the
        one with the largest (254 entity slots; the replay tier of monsoon_amd/fitness.py).
        lanes_per_game selects a hot-kernel variant of the build (0 = default); a value the build does not hold is
        refused."""
        self.lib = _lib.load(extended)
        self.extended = extended
        self.h = ctypes.c_void_p()
        cfg = Config(device, max_games, lanes_per_game, stack_bytes)
        rc = self.lib.monsoon_create(ctypes.byref(cfg), ctypes.byref(self.h))
        if rc != _lib.OK:
            msg = self.lib.monsoon_last_error(self.h if self.h else None)
            if self.h:
                self.lib.monsoon_destroy(self.h)
            self.h = None
            raise MonsoonError(f"monsoon_create failed (status {rc}): {msg.decode() if msg else ''}")
        self.max_games = max_games
        self.device = device
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.monsoon_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _ck(self, rc, what):
        _lib.check(self.h, rc, what, self.lib)

    # ---- Seam G, batched ------------------------------------------------------------------
    def reset(self, seeds, decks, factions=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        n = len(seeds)
        decks = np.ascontiguousarray(decks, dtype=np.uint8)
        if decks.shape == (2, 12):
            decks = np.broadcast_to(decks, (n, 2, 12)).copy()
        if decks.shape != (n, 2, 12):
            raise ValueError(f"decks must be [n][2][12], got {decks.shape}")
        if factions is not None:
            factions = np.ascontiguousarray(factions, dtype=np.uint8).reshape(n, 2)
        self._ck(self.lib.monsoon_reset(self.h, n, _ptr(seeds), _ptr(decks), _ptr(factions)), "monsoon_reset")
        self.n = n

    def legal_mask(self):
        out = np.zeros((self.n, 3), dtype=np.uint64)
        self._ck(self.lib.monsoon_legal_mask(self.h, _ptr(out)), "monsoon_legal_mask")
        return out

    def legal_actions(self, i=0, mask=None):
        m = self.legal_mask()[i] if mask is None else mask
        return [a for a in range(156) if (int(m[a >> 6]) >> (a & 63)) & 1]

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.uint8)
        if actions.shape != (self.n,):
            raise ValueError("one action per game (255 = skip)")
        reward = np.zeros(self.n, dtype=np.int8)
        done = np.zeros(self.n, dtype=np.uint8)
        fault = np.zeros(self.n, dtype=np.uint8)
        self._ck(self.lib.monsoon_step(self.h, _ptr(actions), _ptr(reward), _ptr(done), _ptr(fault)), "monsoon_step")
        return reward, done, fault

    def expert_action(self):
        """Stormbound.expert_action for every game (consumes the games' streams)."""
        action = np.zeros(self.n, dtype=np.uint8)
        fault = np.zeros(self.n, dtype=np.uint8)
        self._ck(self.lib.monsoon_expert_action(self.h, _ptr(action), _ptr(fault)), "monsoon_expert_action")
        return action, fault

    def observe(self):
        out = np.zeros((self.n, 27, 5, 4), dtype=np.int32)
        raises = np.zeros(self.n, dtype=np.uint8)
        self._ck(self.lib.monsoon_observe(self.h, _ptr(out), _ptr(raises)), "monsoon_observe")
        return out, raises

    def game_faults(self):
        """Per-game fault code of the loaded batch (0 = none)."""
        out = np.zeros(self.n, dtype=np.uint8)
        self._ck(self.lib.monsoon_game_faults(self.h, _ptr(out)), "monsoon_game_faults")
        return out

    def observe_torch(self):
        """(n,27,5,4) int32 observation as a torch tensor ON THE GPU (no host copy), plus the raises mask.
        torch is used for device memory only."""
        import torch
        dev = torch.device("cuda", self.device)
        out = torch.empty((self.n, 27, 5, 4), dtype=torch.int32, device=dev)
        raises = torch.empty((self.n,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        self._ck(self.lib.monsoon_observe_dev(self.h, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(raises.data_ptr())),
                 "monsoon_observe_dev")
        return out, raises

    def features(self):
        out = np.zeros((self.n, 10), dtype=np.float64)
        self._ck(self.lib.monsoon_features(self.h, _ptr(out)), "monsoon_features")
        return out

    def status(self):
        out = np.zeros((self.n, 4), dtype=np.int32)
        self._ck(self.lib.monsoon_status(self.h, _ptr(out)), "monsoon_status")
        return out

    def export(self, i):
        buf = np.zeros(2048, dtype=np.uint8)
        ln = ctypes.c_int32()
        self._ck(self.lib.monsoon_state_export(self.h, i, _ptr(buf), ctypes.byref(ln)), "monsoon_state_export")
        return buf[:ln.value].tobytes()

    def variant(self):
        """(candidate lanes per game, waves per SIMD) of the hot kernel this handle runs."""
        u, w = ctypes.c_int32(), ctypes.c_int32()
        self._ck(self.lib.monsoon_variant(self.h, ctypes.byref(u), ctypes.byref(w)), "monsoon_variant")
        return u.value, w.value

    def save_state(self, i):
        """copy.deepcopy of game i (evo/game_adapter.py:280-287): the complete device state as an opaque blob."""
        buf = np.zeros(self.lib.monsoon_state_blob_bytes(), dtype=np.uint8)
        self._ck(self.lib.monsoon_state_save(self.h, i, _ptr(buf), len(buf)), "monsoon_state_save")
        return buf.tobytes()

    def load_state(self, i, blob):
        """Put a blob from save_state (any handle of the same build) into slot i; i == n appends a game."""
        buf = np.frombuffer(blob, dtype=np.uint8)
        self._ck(self.lib.monsoon_state_load(self.h, i, _ptr(buf), len(buf)), "monsoon_state_load")
        if i == self.n:
            self.n = i + 1

    def debug_build(self, i, seed, stream_pos, state):
        """Scenario tests: put game i into a described state (int32 stream, tests/scenario_lib.py); returns the fault code."""
        state = np.ascontiguousarray(state, dtype=np.int32)
        f = ctypes.c_int32()
        self._ck(self.lib.monsoon_debug_build(self.h, i, int(seed) & 0xFFFFFFFF, int(stream_pos), _ptr(state), len(state), ctypes.byref(f)),
                 "monsoon_debug_build")
        if i == self.n:
            self.n = i + 1
        return f.value

    def debug_op(self, i, op):
        """One engine call on game i; returns (fault code, [[card, position], ...] of the abilities that ran, in order)."""
        op = np.ascontiguousarray(op, dtype=np.int32)
        log = np.zeros(512, dtype=np.int32)
        f, n = ctypes.c_int32(), ctypes.c_int32()
        self._ck(self.lib.monsoon_debug_op(self.h, i, _ptr(op), len(op), ctypes.byref(f), _ptr(log), 256, ctypes.byref(n)), "monsoon_debug_op")
        return f.value, log[:2 * n.value].reshape(-1, 2).tolist()

    def debug_kat(self, kind, seed, n, inp=None):
        """Known-answer diagnostics (monsoon_debug_kat): 0 raw u32, 1 random(), 2 randint(0, inp[i]), 3 shuffle of range(12),
        4 scores of inp[n][30] = {weights, before, after}."""
        out = np.zeros((n, 12) if kind == 3 else n, dtype={0: np.uint32, 1: np.float64, 2: np.int32, 3: np.int32, 4: np.float64}[kind])
        if inp is not None:
            inp = np.ascontiguousarray(inp, dtype=np.int32 if kind == 2 else np.float64)
        self._ck(self.lib.monsoon_debug_kat(self.h, kind, int(seed) & 0xFFFFFFFF, n, _ptr(inp), _ptr(out)), "monsoon_debug_kat")
        return out

    def debug_raw(self, i):
        buf = np.zeros(4096, dtype=np.uint8)
        ln = ctypes.c_int32()
        self._ck(self.lib.monsoon_debug_raw(self.h, i, _ptr(buf), ctypes.byref(ln)), "monsoon_debug_raw")
        return buf[:ln.value].copy()

    def state_hash(self):
        out = np.zeros(self.n, dtype=np.uint64)
        self._ck(self.lib.monsoon_state_hash(self.h, _ptr(out)), "monsoon_state_hash")
        return out

    # ---- Seam F ---------------------------------------------------------------------------
    def decide(self, weights, want_scores=False):
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if weights.shape == (10,):
            weights = np.broadcast_to(weights, (self.n, 2, 10)).copy()
        if weights.shape != (self.n, 2, 10):
            raise ValueError("weights must be [n][2][10]")
        action = np.zeros(self.n, dtype=np.uint8)
        best = np.zeros(self.n, dtype=np.float64)
        scores = np.zeros((self.n, 156), dtype=np.float64) if want_scores else None
        self._ck(self.lib.monsoon_decide(self.h, _ptr(weights), _ptr(action), _ptr(best), _ptr(scores)), "monsoon_decide")
        return (action, best, scores) if want_scores else (action, best)

    def rollout(self, weights, matches, deck_pairs, max_turns, want_results=False):
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        n_ind = weights.shape[0]
        deck_pairs = np.ascontiguousarray(deck_pairs, dtype=np.uint8).reshape(-1, 2, 12)
        m = np.ascontiguousarray(matches)
        if m.dtype != np.dtype([("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")]):
            arr = np.zeros(len(matches), dtype=[("p1", "<i4"), ("p2", "<i4"), ("seed", "<u4"), ("deck", "<u4")])
            mm = np.asarray(matches)
            arr["p1"], arr["p2"], arr["seed"], arr["deck"] = mm[:, 0], mm[:, 1], mm[:, 2], mm[:, 3]
            m = arr
        counts = np.zeros((n_ind, 3), dtype=np.int32)
        results = np.zeros(len(m), dtype=np.int8) if want_results else None
        steps = np.zeros(len(m), dtype=np.int32) if want_results else None
        self._ck(self.lib.monsoon_rollout(self.h, _ptr(weights), n_ind, _ptr(m), len(m), _ptr(deck_pairs), len(deck_pairs),
                                          max_turns, _ptr(counts), _ptr(results), _ptr(steps)), "monsoon_rollout")
        # the handle now holds the last batch of the schedule
        self.n = len(m) - ((len(m) - 1) // self.max_games) * self.max_games
        return (counts, results, steps) if want_results else counts

    def rollout_faults(self, n_matches):
        """Fault code of every game of the last rollout (>= 16: a limit of this build's record, see include/monsoon.h)."""
        out = np.zeros(n_matches, dtype=np.uint8)
        self._ck(self.lib.monsoon_rollout_faults(self.h, _ptr(out), n_matches), "monsoon_rollout_faults")
        return out

    def draw_decks(self, seeds, pool):
        """uint8[n][2][12]: RandomState(seed).choice(pool, 12, replace=False) twice per seed, drawn on the device
        (monsoon_draw_decks; configuration C5's per-game decks)."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        out = np.zeros((len(seeds), 2, 12), dtype=np.uint8)
        if len(seeds):
            self._ck(self.lib.monsoon_draw_decks(self.h, _ptr(seeds), len(seeds), _ptr(pool), len(pool), _ptr(out)), "monsoon_draw_decks")
        return out

    # ---- GA operators on the device (the host GA stays the default) -------------------------
    def ga_offspring(self, np_state, parents_w, parents_s, n_offspring, tau, tau_prime, min_sigma):
        """Population.generate_offspring on the device over numpy's global stream.  np_state: RandomState.get_state()
        (legacy tuple or dict).  Returns (weights[n][dim], sigmas[n][dim], parent[n], tries[n], new_state) with new_state
        in the form np.random.set_state accepts."""
        from ._lib import NpState
        if isinstance(np_state, dict):
            key, pos, hg, g = np_state["state"]["key"], np_state["state"]["pos"], np_state["has_gauss"], np_state["gauss"]
        else:
            _, key, pos, hg, g = np_state
        st = NpState()
        key = np.ascontiguousarray(key, dtype=np.uint32)   # kept alive across the memmove
        ctypes.memmove(st.key, key.ctypes.data, 624 * 4)
        st.pos, st.has_gauss, st.gauss = int(pos), int(hg), float(g)
        pw = np.ascontiguousarray(parents_w, dtype=np.float64)
        ps = np.ascontiguousarray(parents_s, dtype=np.float64)
        mu, dim = pw.shape
        ow, osg = np.zeros((n_offspring, dim)), np.zeros((n_offspring, dim))
        par, tries = np.zeros(n_offspring, dtype=np.int32), np.zeros(n_offspring, dtype=np.int64)
        self._ck(self.lib.monsoon_ga_offspring(self.h, ctypes.byref(st), _ptr(pw), _ptr(ps), mu, dim, n_offspring, float(tau), float(tau_prime),
                                               float(min_sigma), _ptr(ow), _ptr(osg), _ptr(par), _ptr(tries)), "monsoon_ga_offspring")
        new_key = np.ctypeslib.as_array(st.key).astype(np.uint32).copy()
        return ow, osg, par, tries, ("MT19937", new_key, int(st.pos), int(st.has_gauss), float(st.gauss))

    def ga_select(self, fitness):
        """Indices of all individuals by descending fitness, ties in original order (select_from_combined's sort)."""
        f = np.ascontiguousarray(fitness, dtype=np.float64)
        out = np.zeros(len(f), dtype=np.int32)
        self._ck(self.lib.monsoon_ga_select(self.h, _ptr(f), len(f), _ptr(out)), "monsoon_ga_select")
        return out

    # ---- device-resident rounds (bench) -------------------------------------------------------
    def upload_weights(self, weights):
        weights = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1, 10)
        self._ck(self.lib.monsoon_upload_weights(self.h, _ptr(weights), len(weights)), "monsoon_upload_weights")

    def assign_players(self, p1, p2):
        p1 = np.ascontiguousarray(p1, dtype=np.int32)
        p2 = np.ascontiguousarray(p2, dtype=np.int32)
        self._ck(self.lib.monsoon_assign_players(self.h, _ptr(p1), _ptr(p2)), "monsoon_assign_players")

    def decide_round(self):
        self._ck(self.lib.monsoon_decide_round_dev(self.h), "monsoon_decide_round_dev")

    def play_rounds(self, rounds):
        """`rounds` decisions of every loaded game in one launch (asynchronous; sync() waits)."""
        self._ck(self.lib.monsoon_play_rounds_dev(self.h, rounds), "monsoon_play_rounds_dev")

    def sync(self):
        self._ck(self.lib.monsoon_sync(self.h), "monsoon_sync")

    def stats(self):
        s = Stats()
        self._ck(self.lib.monsoon_get_stats(self.h, ctypes.byref(s)), "monsoon_get_stats")
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    def reset_stats(self):
        self._ck(self.lib.monsoon_reset_stats(self.h), "monsoon_reset_stats")

    def kernel_time(self):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        self._ck(self.lib.monsoon_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(n)), "monsoon_kernel_time")
        return ms.value, n.value
