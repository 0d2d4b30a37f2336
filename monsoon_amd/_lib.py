"""ctypes binding of libmonsoon_hip.so (include/monsoon.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C monsoon_amd/csrc`.
There is no fallback: if the shared object is missing or no gfx950 device is usable, every
entry point of this package raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmonsoon_hip.so")
# same source built with -DMSB_EXT=1: larger per-game record, needed by decks holding ua20 or b005
LIB_PATH_EXT = os.path.join(_HERE, "libmonsoon_hip_ext.so")
# -DMSB_EXT=2: the large extended record (254 entity slots): the games whose nested b005 memories outgrow the extended
# record are replayed here (monsoon_amd/fitness.py)
LIB_PATH_BIG = os.path.join(_HERE, "libmonsoon_hip_big.so")

OK, ERR_ARG, ERR_DEVICE, ERR_STATE = 0, 1, 2, 3
NUM_ACTIONS = 156
OBS_INTS = 540


class Config(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("max_games", ctypes.c_int32), ("lanes_per_game", ctypes.c_int32),
                ("stack_bytes", ctypes.c_int32)]


class Match(ctypes.Structure):
    _fields_ = [("p1", ctypes.c_int32), ("p2", ctypes.c_int32), ("seed", ctypes.c_uint32), ("deck", ctypes.c_uint32)]


class Stats(ctypes.Structure):
    _fields_ = [("lookahead_steps", ctypes.c_uint64), ("decisions", ctypes.c_uint64), ("games_finished", ctypes.c_uint64),
                ("faults", ctypes.c_uint64), ("capacity_faults", ctypes.c_uint64), ("lookahead_capacity_faults", ctypes.c_uint64)]


class MonsoonError(RuntimeError):
    pass


_libs = {}

# name -> (restype, argtypes); every symbol declared in include/monsoon.h
class NpState(ctypes.Structure):   # monsoon_np_state: numpy.random.RandomState.get_state() of the global stream
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32), ("gauss", ctypes.c_double)]


SIGNATURES = {
    "monsoon_create": (ctypes.c_int, [ctypes.POINTER(Config), ctypes.POINTER(ctypes.c_void_p)]),
    "monsoon_destroy": (None, [ctypes.c_void_p]),
    "monsoon_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "monsoon_version": (ctypes.c_int, []),
    "monsoon_variant": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "monsoon_state_blob_bytes": (ctypes.c_int32, []),
    "monsoon_state_save": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]),
    "monsoon_state_load": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]),
    "monsoon_card_index": (ctypes.c_int, [ctypes.c_char_p]),
    "monsoon_card_supported": (ctypes.c_int, [ctypes.c_int]),
    "monsoon_reset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_legal_mask": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_expert_action": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_observe": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_game_faults": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_observe_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_features": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_state_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "monsoon_debug_build": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int32,
                                           ctypes.POINTER(ctypes.c_int32)]),
    "monsoon_debug_op": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                        ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    "monsoon_debug_kat": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_debug_raw": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "monsoon_state_hash": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_decide": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_rollout": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                       ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "monsoon_rollout_faults": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "monsoon_draw_decks": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "monsoon_ga_offspring": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_ga_select": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "monsoon_upload_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "monsoon_assign_players": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_decide_round_dev": (ctypes.c_int, [ctypes.c_void_p]),
    "monsoon_play_rounds_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "monsoon_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "monsoon_get_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Stats)]),
    "monsoon_debug_counters": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "monsoon_reset_stats": (ctypes.c_int, [ctypes.c_void_p]),
    "monsoon_kernel_time": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "monsoon_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
}


def load(extended=False):
    """Load the HIP library (fails loudly when it has not been built).  extended: False / True / 2 (the large record)."""
    extended = int(extended)
    if extended in _libs:
        return _libs[extended]
    path = (LIB_PATH, LIB_PATH_EXT, LIB_PATH_BIG)[extended]
    if not extended and os.environ.get("MONSOON_LIB"):
        path = os.environ["MONSOON_LIB"]   # development knob: A/B another build of the same ABI (scripts/ab_libs.sh)
    if not os.path.exists(path):
        raise MonsoonError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C monsoon_amd/csrc all` (there is no CPU fallback)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        if path not in (LIB_PATH, LIB_PATH_EXT, LIB_PATH_BIG) and not hasattr(lib, name):
            continue              # an older build loaded through MONSOON_LIB may lack newer diagnostics entry points
        fn = getattr(lib, name)   # AttributeError if the ABI and this binding drift apart
        fn.restype = res
        fn.argtypes = args
    _libs[extended] = lib
    return lib


def check(handle, rc, what, lib=None):
    if rc != OK:
        msg = (lib or load()).monsoon_last_error(handle)
        raise MonsoonError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
