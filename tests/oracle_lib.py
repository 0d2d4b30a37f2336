"""ctypes wrapper of oracle/liboracle.so -- the CPU replay oracle (test infrastructure).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
LIB_PATH_EXT = os.path.join(ORACLE_DIR, "liboracle_ext.so")   # extended record (ua20, b005)
LIB_PATH_BIG = os.path.join(ORACLE_DIR, "liboracle_big.so")   # large extended record (extended=2)
# host builds of the PRODUCT's rules core (explicit work stack; oracle/oracle.cpp -DORC_PRODUCT_CORE): the same C entry
# points over the other implementation of the rules, for the CPU-side checks of the product core
PRODUCT_HOST = tuple(os.path.join(ORACLE_DIR, n) for n in ("libproduct_host.so", "libproduct_host_ext.so", "libproduct_host_big.so"))
# ... and with the device's 21-word resident work stack, so that the eviction path of wk_reserve runs on the CPU too
PRODUCT_HOST_EVICT = tuple(os.path.join(ORACLE_DIR, n) for n in ("libproduct_host_evict.so", "libproduct_host_evict_ext.so"))


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "all"], check=True)


_libs = {}


def lib(extended=False, core=None):
    core = core or os.environ.get("MSB_ORACLE_CORE", "oracle")   # "product": the host build of the product's rules core
    key = (extended, core)
    if key not in _libs:
        paths = {"product": PRODUCT_HOST, "product_evict": PRODUCT_HOST_EVICT}.get(core, (LIB_PATH, LIB_PATH_EXT, LIB_PATH_BIG))
        path = extended if isinstance(extended, str) else paths[int(extended)]
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_reset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.orc_legal.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.orc_expert_action.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.orc_observe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_features.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_canon.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_canon_hash.restype = ctypes.c_uint64
        L.orc_canon_hash.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_obs_hash.restype = ctypes.c_uint64
        L.orc_obs_hash.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_have_winner.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_to_play.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_decide.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_lookahead_faults.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_lookahead_faults.restype = None
        L.orc_game_fault.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_rollout.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64),
                                  ctypes.POINTER(ctypes.c_int)]
        L.orc_rollout_batch.restype = ctypes.c_uint64
        L.orc_rollout_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_score.restype = ctypes.c_double
        L.orc_score.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_rng_u32.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        L.orc_rng_random.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        L.orc_rng_randint.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_rng_shuffle.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fnv1a64(data):
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


class Oracle:
    """n independent games replayed on the CPU."""

    def __init__(self, n=1, extended=False, core=None):
        # core: "oracle" (the recursive restatement, the checker) or "product" (host build of the product's rules core);
        # MSB_ORACLE_CORE=product switches the default, which runs the whole golden suite over the product core
        self.L = lib(extended, core)
        self.h = ctypes.c_void_p(self.L.orc_create(n))
        self.n = n

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass

    def reset(self, i, seed, deck0, deck1, f0=0, f1=0):
        d0 = np.ascontiguousarray(deck0, dtype=np.uint8)
        d1 = np.ascontiguousarray(deck1, dtype=np.uint8)
        return self.L.orc_reset(self.h, i, int(seed) & 0xFFFFFFFF, _p(d0), _p(d1), f0, f1)

    def legal_mask(self, i):
        m = np.zeros(3, dtype=np.uint64)
        self.L.orc_legal(self.h, i, _p(m))
        return m

    def legal_actions(self, i):
        m = self.legal_mask(i)
        return [a for a in range(156) if (int(m[a >> 6]) >> (a & 63)) & 1]

    def step(self, i, action):
        r, d = ctypes.c_int(), ctypes.c_int()
        f = self.L.orc_step(self.h, i, int(action), ctypes.byref(r), ctypes.byref(d))
        return f, r.value, d.value

    def expert_action(self, i):
        f = ctypes.c_int()
        a = self.L.orc_expert_action(self.h, i, ctypes.byref(f))
        return a, f.value

    def observe(self, i):
        out = np.zeros(540, dtype=np.int32)
        if self.L.orc_observe(self.h, i, _p(out)):
            return None
        return out.reshape(27, 5, 4)

    def features(self, i):
        f = np.zeros(10)
        if self.L.orc_features(self.h, i, _p(f)):
            return None
        return f

    def canon(self, i):
        buf = np.zeros(2048, dtype=np.uint8)
        n = self.L.orc_canon(self.h, i, _p(buf))
        return buf[:n].tobytes()

    def canon_hash(self, i):
        return int(self.L.orc_canon_hash(self.h, i))

    def obs_hash(self, i):
        return int(self.L.orc_obs_hash(self.h, i))

    def have_winner(self, i):
        return bool(self.L.orc_have_winner(self.h, i))

    def to_play(self, i):
        return self.L.orc_to_play(self.h, i)

    def decide(self, i, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        scores = np.zeros(156)
        mask = np.zeros(3, dtype=np.uint64)
        a = self.L.orc_decide(self.h, i, _p(w), _p(scores), _p(mask))
        return a, scores, mask

    def scn_build(self, i, seed, stream_pos, state):
        """Scenario tests: game i from a state stream (tests/scenario_lib.py), stream = RandomState(seed) + stream_pos."""
        state = np.ascontiguousarray(state, dtype=np.int32)
        return self.L.orc_scn_build(self.h, i, ctypes.c_uint32(int(seed)), ctypes.c_uint32(int(stream_pos)), _p(state))

    def scn_op(self, i, op):
        """One recorded engine call; returns (fault code, [[card, position tile or -1], ...] of the abilities that ran)."""
        op = np.ascontiguousarray(op, dtype=np.int32)
        log = np.zeros(2 * 256, dtype=np.int32)
        n = ctypes.c_int()
        f = self.L.orc_scn_op(self.h, i, _p(op), _p(log), 256, ctypes.byref(n))
        return f, log[:2 * n.value].reshape(-1, 2).tolist()

    def lookahead_faults(self, i):
        """Fault code of each legal action's look-ahead (255 = not legal); 20 = flagged as unsupported by this build."""
        out = np.zeros(156, dtype=np.uint8)
        self.L.orc_lookahead_faults(self.h, i, _p(out))
        return out

    def rollout(self, i, w1, w2, max_turns, trace=False):
        w1 = np.ascontiguousarray(w1, dtype=np.float64)
        w2 = np.ascontiguousarray(w2, dtype=np.float64)
        acts = np.zeros(max_turns, dtype=np.uint8) if trace else None
        hashes = np.zeros(max_turns, dtype=np.uint64) if trace else None
        ns, nl, fl = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_int()
        res = self.L.orc_rollout(self.h, i, _p(w1), _p(w2), max_turns, _p(acts) if trace else None,
                                 _p(hashes) if trace else None, ctypes.byref(ns), ctypes.byref(nl), ctypes.byref(fl))
        out = dict(result=res, steps=ns.value, lookahead=nl.value, fault=fl.value)
        if trace:
            out["actions"] = acts[:ns.value]
            out["hashes"] = hashes[:ns.value]
        return out

    def game_fault(self, i):
        """monsoon_game_faults of the product: the fault that stopped game i, else the first capacity code a look-ahead hit."""
        return int(self.L.orc_game_fault(self.h, i))

    def rollout_batch(self, n, w, max_turns, threads):
        """Rollouts of games [0, n) (same weights both sides) on `threads` host threads."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        results = np.zeros(n, dtype=np.int8)
        steps = np.zeros(n, dtype=np.int32)
        hashes = np.zeros(n, dtype=np.uint64)
        total = self.L.orc_rollout_batch(self.h, n, _p(w), max_turns, threads, _p(results), _p(steps), _p(hashes))
        return int(total), results, steps, hashes
