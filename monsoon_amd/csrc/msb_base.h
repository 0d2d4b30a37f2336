// Monsoon-AMD: shared base definitions for the batched Stormbound rules core.
//
// The rules core (rules.h, features.h) is ONE source compiled twice:
//   * by hipcc for gfx950 inside the product kernels (monsoon_hip.hip), where
//     every state access goes through a lane-interleaved LDS accessor, and
//   * by g++ for the host inside oracle/ (test infrastructure only), where the
//     same accessor interface is backed by a plain byte array.
// Nothing in the product library calls the host build.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MSB_HD __device__
#define MSB_NOINLINE __attribute__((noinline))
#else
#define MSB_HD
#define MSB_NOINLINE __attribute__((noinline))
#endif
#define MSB_INL inline __attribute__((always_inline))

// Function-level timing scopes of the profiling build (-DMSB_PROF=1, scripts/phase_profile.py only): the leader
// lane of whatever sub-wave executes a function adds the elapsed wave cycles to an LDS counter.  Inclusive
// times; nested/recursive scopes count twice.  Expands to nothing in the product and in the oracle.
#define MSB_PROF_LDS 16          // u64 cycles[32], u32 calls[32], u64 entry[32], u64 exit[32], two u64 stamps; records start at 928
#if defined(MSB_PROF) && MSB_PROF && defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) unsigned long long* msb_lds64;
typedef __attribute__((address_space(3))) unsigned* msb_lds32;
__device__ inline __attribute__((always_inline)) bool msb_prof_leader() {
  unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  return (unsigned)__builtin_amdgcn_readfirstlane((int)l) == l;
}
struct ProfScope {
  unsigned long long t0;
  int id;
  __device__ inline __attribute__((always_inline)) explicit ProfScope(int i) : t0(__builtin_readcyclecounter()), id(i) {
    // time since the caller's MSB_PRECALL stamp = call + prologue (callee-saved register saves)
    if (msb_prof_leader()) {
      unsigned long long pre = *(msb_lds64)(unsigned long)(MSB_PROF_LDS + 896);
      if (pre) {
        __hip_atomic_fetch_add((msb_lds64)(unsigned long)(MSB_PROF_LDS + 384 + 8 * id), t0 - pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        *(msb_lds64)(unsigned long)(MSB_PROF_LDS + 896) = 0ull;
      }
    }
  }
  __device__ inline __attribute__((always_inline)) ~ProfScope() {
    unsigned long long now = __builtin_readcyclecounter();
    if (msb_prof_leader()) {
      __hip_atomic_fetch_add((msb_lds64)(unsigned long)(MSB_PROF_LDS + 8 * id), now - t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add((msb_lds32)(unsigned long)(MSB_PROF_LDS + 256 + 4 * id), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      *(msb_lds64)(unsigned long)(MSB_PROF_LDS + 904) = now;   // the epilogue starts here
    }
  }
};
#define MSB_SCOPE(id) ProfScope msb_prof_scope_(id)
#define MSB_PRECALL()                                                                                   \
  do {                                                                                                  \
    if (msb_prof_leader()) *(msb_lds64)(unsigned long)(MSB_PROF_LDS + 896) = __builtin_readcyclecounter(); \
  } while (0)
#define MSB_POSTCALL(id)                                                                                \
  do {                                                                                                  \
    unsigned long long now_ = __builtin_readcyclecounter();                                             \
    if (msb_prof_leader()) {                                                                            \
      unsigned long long post_ = *(msb_lds64)(unsigned long)(MSB_PROF_LDS + 904);                       \
      if (post_ && now_ > post_)                                                                        \
        __hip_atomic_fetch_add((msb_lds64)(unsigned long)(MSB_PROF_LDS + 640 + 8 * (id)), now_ - post_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      *(msb_lds64)(unsigned long)(MSB_PROF_LDS + 904) = 0ull;                                           \
    }                                                                                                   \
  } while (0)
#else
#define MSB_SCOPE(id)
#define MSB_PRECALL() do {} while (0)
#define MSB_POSTCALL(id) do {} while (0)
#endif
enum {
  PS_STEP, PS_PLAYER_PLAY, PS_NEW_ENTITY, PS_RUN_ABILITY, PS_ABILITY_ENTITY, PS_ABILITY_SPELL, PS_GET_TARGETS, PS_SHAPE_TILES,
  PS_SHAPE_TARGETS, PS_DEAL_DAMAGE, PS_DESTROY, PS_FRONT_LINE, PS_SET_PATH, PS_MOVE, PS_COMMAND, PS_FORCE_ATTACK, PS_DRAW,
  PS_FLIP, PS_NEXT_TURN, PS_LEGAL, PS_SHUFFLE, PS_SORTED_HEAD, PS_SPAWN, PS_RESPAWN, PS_TELEPORT, PS_PUSH_PULL, PS_EMPTY_FRONT,
  PS_BEGIN_STEP, PS_OBS_RAISES, PS_FEATURES, PS_COUNT
};


namespace msb {

// ---- enums (reference enums.py) -------------------------------------------------------
enum : int { KIND_UNIT = 0, KIND_STRUCT = 1, KIND_SPELL = 2 };
// TriggerType, enums.py:69-79
enum : int { TR_NONE = -1, TR_ON_PLAY = 0, TR_ON_DEATH = 1, TR_BEFORE_ATTACKING = 2, TR_AFTER_ATTACKING = 3,
             TR_AFTER_SURVIVING = 4, TR_BEFORE_MOVING = 5, TR_TURN_START = 6, TR_TURN_END = 7 };
// StatusEffect, enums.py:81-86
enum : int { ST_FROZEN = 0, ST_POISONED = 1, ST_CONFUSED = 2, ST_DISABLED = 3, ST_VITALIZED = 4 };
// Phase, enums.py:88-91
enum : int { PH_TURN_START = 0, PH_PLAY = 1, PH_TURN_END = 2 };
// UnitType, enums.py:51-67
enum : int { UT_CONSTRUCT = 0, UT_FLAKE, UT_KNIGHT, UT_PIRATE, UT_RAVEN, UT_RODENT, UT_SATYR, UT_TOAD, UT_UNDEAD,
             UT_VIKING, UT_HERO, UT_DRAGON, UT_ELDER, UT_FELINE, UT_ANCIENT, UT_PRIMAL };
// Target.Kind / Target.Side, target.py:8-16
enum : int { TK_UNIT = 0, TK_STRUCTURE = 1, TK_ANY = 2 };
enum : int { TS_FRIENDLY = 0, TS_ENEMY = 1, TS_ANY = 2 };

// Per-game fault codes: the reference raises Python exceptions that its agent layer swallows
// (evo/heuristic_agent.py:48-51, evo/fitness.py:208-210); here a step that would raise sets a
// fault byte instead.  Codes >= FAULT_CAPACITY are build limits, not reference behaviour; the
// tests require them to be zero on every benchmark configuration.
enum : int {
  FAULT_NONE = 0,
  FAULT_PY_EXCEPTION = 1,   // reference raises (u310/u017/s101, Point.__eq__(None), ...)
  FAULT_INT_CARD = 2,       // int(card) ValueError for up01/up02/up03 (card.py:46)
  FAULT_CAPACITY = 16,      // entity slots / deck / path capacity
  FAULT_TRIG_STACK = 17,    // deferred-trigger stack overflow
  FAULT_DEPTH = 18,         // recursion guard
  FAULT_RNG_OVERRUN = 19,   // one step consumed more than the two resident MT blocks
  FAULT_UNSUPPORTED = 20,   // card ability not implemented yet in this build
  FAULT_STATUS_SAT = 21,    // status multiset count saturated
  FAULT_CAP_REM = 22,       // b005 snapshot lists
  FAULT_CAP_DECK = 23,      // deck entries
  FAULT_CAP_HAND = 24,      // hand entries
  FAULT_CAP_PATH = 25,      // path longer than PATH_CAP
  FAULT_CAP_INST = 26,      // card-instance strength outside 0..255
};

// ---- static card table -----------------------------------------------------------------
struct TargetSpec {
  int8_t has, kind, side;
  uint16_t types, xtypes;
  int16_t limit;  // -1 = None
  int8_t non_hero;
  uint8_t status, xstatus;
  int8_t base;
};
struct CardInfo {
  int8_t kind, faction;
  uint8_t cost, strength, movement;
  int8_t trigger, ff, has_ability;
  uint16_t types;
  int8_t first_type;
  int32_t int_id;
  TargetSpec tgt;
};

constexpr int NUM_CARDS = 112;
constexpr int TOKEN_UNIT_BASE = 112;   // + UnitType
constexpr int TOKEN_STRUCT = 128;
constexpr int CARD_NONE = 0xFF;

#if defined(__HIPCC__)
__device__ __constant__
#endif
static const CardInfo g_cards[NUM_CARDS] = {
#include "card_table.inc"
};

// Card.weight values: wtab[k] = f^k(1), f(w) = w * 1.6 + 100 with both operations rounded separately, exactly as the
// reference's Python floats compute them (player.py:31,57-59).  Constant-evaluated: no contraction, round to nearest.
struct WeightTable {
  double v[256];
};
constexpr WeightTable make_weight_table() {
  WeightTable t{};
  double w = 1.0;
  for (int i = 0; i < 256; i++) {
    t.v[i] = w;
    double m = w * 1.6;
    w = m + 100.0;
  }
  return t;
}
#if defined(__HIPCC__)
__device__ __constant__
#endif
static const WeightTable g_wtab = make_weight_table();

// Card indices used by name in rules.h (sorted-id order; checked against card_ids.json by tests).
#include "card_names.inc"

}  // namespace msb
