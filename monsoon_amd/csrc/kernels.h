// kernels.h -- device code shared by every translation unit of libmonsoon_hip.so: per-game buffers, stream-block
// maintenance, and the hot kernel k_play as a template over {candidate lanes per game U, waves per SIMD W} (the builds'
// default is its sibling k_play_reg, kernels_reg.h: the same decision loop with the game's record in registers).  Each
// instantiation is a full compilation of the rules core, so every variant lives in a translation unit of its own
// (variant.hip, built in parallel by the Makefile); monsoon_hip.hip holds the API kernels and the host side.
//
// Execution model
//   * Hot kernel k_play<U,W>: ONE WAVEFRONT PER GAME at a time (a persistent grid of resident wavefronts, each popping
//     game indices from its range's counter).  The game's record is staged from HBM into LDS once and stays there
//     for up to `rounds` decisions.  Per decision: the legal-action mask is evaluated on that shared copy (LDS
//     broadcast reads); then up to U candidate actions are advanced at once, lane l stepping its own private copy
//     of the state.  The private copies are interleaved across lanes in 16-byte granules (granule c of lane l at
//     (c*U + l)*16), so lanes touching the same field hit distinct LDS banks and a whole entity is one
//     ds_read_b128.  Scores are reduced with shuffles over the U candidate lanes (first maximum in ascending action
//     order = np.argmax over the sorted legal list) and the winner's column becomes the game's record.  Nothing
//     is re-executed: the committed successor IS one of the look-ahead results (when the legal set needs several
//     passes of U lanes, the best successor so far is parked in the game's HBM record), and its features are the next
//     decision's "before" features.
//   * The game's MT19937 stream lives in HBM as two blocks of tempered outputs (current + next) plus the raw
//     state; candidate steps read it through a private cursor, the committed cursor travels with the record and
//     the wave regenerates a block (twist in LDS) when it is used up.
//   * Integer/index work: no MFMA.  f64 appears only in the weighted draw and the score.
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include "../../include/monsoon.h"
#include "rules.h"
#include "canon.h"

namespace msbk {
using namespace msb;


constexpr int SW = STATE_WORDS;               // record stride in HBM, words (STATE_BYTES is a multiple of 16)
constexpr int SG = STATE_BYTES / 16;         // 16-byte granules per record
static_assert(STATE_BYTES % 16 == 0, "record must be whole granules");
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int RNG_WORDS = 2 * MT_N;          // tempered outputs: two blocks per game

struct GameMeta {
  int32_t p1, p2;          // weight-table rows of the FIRST / SECOND player
  int8_t result;           // -2 running, -1 draw, 0 FIRST won, 1 SECOND won
  uint8_t fault;
  uint8_t last_action;
  uint8_t flags;           // b0: ended with a winner (have_winner), as opposed to max_turns / a fault
  uint16_t steps;          // committed steps (decisions and monsoon_step calls)
  uint16_t decided;        // decisions committed by k_decide
  uint32_t rng;            // cursor (bits 0-15) | current block (bit 16)
  uint32_t lookahead;      // look-ahead transitions executed for this game
  uint32_t match;          // schedule index (rollout)
  uint8_t la_fault;        // first build-limit fault (code >= 16) a LOOK-AHEAD of this game hit: that action was scored
                           // 0.0 where the reference computes a score, so the game may have left the reference's line
  uint8_t pad_[3];
};
static_assert(sizeof(GameMeta) == 32, "one 32-byte row per game");

struct DevBuffers {
  uint32_t* state;     // [cap][SW]
  uint32_t* rng_out;   // [cap][2][624]
  uint32_t* rng_mt;    // [cap][624]
  GameMeta* meta;      // [cap]
  double* weights;     // [n_individuals][10]
  unsigned long long* stats;  // [8]: lookahead, decisions, finished, faults, capacity_faults, look-ahead capacity faults
  double* scores;      // [cap][156] or null
  double* best;        // [cap]
  int* pop;            // [2][POP_PARTS * POP_STRIDE] game-index counters of the persistent k_decide, alternating between launches
  unsigned long long* prof;   // [cap][..] phase cycles, scope cycles, scope calls (profiling build), profiling build only (else null)
  uint32_t* wk_ovf;    // overflow blocks of the rules core's work stack: [workgroup][SK_CAP - SKW][lanes stepping games in it]
};

// Work-stack words per game kept in LDS (state.h LaneMem, rules.h wk_reserve): 98 % of the steps of a neutral-deck game
// never hold more than 8; a step that wants more than SKW - SK_NEED when it enters an ability or a move parks what it has
// in the workgroup's eviction block in HBM.
#if defined(MSB_SKW)
constexpr int SKW = MSB_SKW;
#else
constexpr int SKW = 21;
#endif
static_assert(SKW >= SK_NEED + 9 && SKW * 4 >= 80, "room for the largest frame + the eviction mark + what a handler pushes; the candidates' features overlay their stacks");
constexpr int OVF_WORDS = SK_CAP;   // per stepping lane

enum { ST_LOOKAHEAD = 0, ST_DECISIONS = 1, ST_FINISHED = 2, ST_FAULTS = 3, ST_CAPFAULTS = 4, ST_LACAPFAULTS = 5, ST_N = 6, ST_PROF = 8, ST_WORDS = 32, PROF_WORDS = 138 };
// Phase timing of k_decide (profiling build only, -DMSB_PROF=1 -> libmonsoon_hip_prof.so; never the product):
// wave cycles per phase accumulated into stats[ST_PROF + phase].
#if defined(MSB_PROF) && MSB_PROF
#define PROF_DECL()                                                                                   \
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                          \
  const unsigned long long prof_wall0 = wall_clock64();                                               \
  for (int i_ = lane; i_ < (928 - 16) / 4; i_ += 64) *(MSB_AS_LDS unsigned*)(uintptr_t)(MSB_PROF_LDS + 4 * i_) = 0u; \
  __syncthreads();                                                                                    \
  unsigned long long prof_t = __builtin_readcyclecounter()
#define PROF_MARK(ph)                                         \
  do {                                                        \
    unsigned long long now_ = __builtin_readcyclecounter();   \
    prof_acc[ph] += now_ - prof_t;                            \
    prof_t = now_;                                            \
  } while (0)
#define PROF_FLUSH()                                                        \
  do {                                                                      \
    __syncthreads();                                                        \
    if (lane == 0)                                                          \
      for (int i_ = 0; i_ < 8; i_++) b.prof[(size_t)g * PROF_WORDS + i_] += prof_acc[i_]; \
    if (lane == 0) {                                                        \
      b.prof[(size_t)g * PROF_WORDS + 136] = prof_wall0;                    \
      b.prof[(size_t)g * PROF_WORDS + 137] = wall_clock64();                \
    }                                                                       \
    if (lane < 32) {                                                        \
      b.prof[(size_t)g * PROF_WORDS + 8 + lane] += *(MSB_AS_LDS unsigned long long*)(uintptr_t)(MSB_PROF_LDS + 8 * lane); \
      b.prof[(size_t)g * PROF_WORDS + 40 + lane] += *(MSB_AS_LDS unsigned*)(uintptr_t)(MSB_PROF_LDS + 256 + 4 * lane);   \
      b.prof[(size_t)g * PROF_WORDS + 72 + lane] += *(MSB_AS_LDS unsigned long long*)(uintptr_t)(MSB_PROF_LDS + 384 + 8 * lane); \
      b.prof[(size_t)g * PROF_WORDS + 104 + lane] += *(MSB_AS_LDS unsigned long long*)(uintptr_t)(MSB_PROF_LDS + 640 + 8 * lane); \
    }                                                                       \
  } while (0)
#else
#define PROF_DECL() do {} while (0)
#define PROF_MARK(ph) do {} while (0)
#define PROF_FLUSH() do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// RNG block maintenance (wave-cooperative, in LDS)
// ------------------------------------------------------------------------------------------------
// In-place MT19937 twist of 624 words in LDS by one wavefront.  Within one pass all lanes read
// before any lane writes (a wave executes in lockstep), and passes are ordered by barriers.
__device__ inline void wave_twist_lds(MSB_AS_LDS uint32_t* mt, int lane) {
  for (int k0 = 0; k0 < MT_N - MT_M; k0 += 64) {
    int k = k0 + lane;
    uint32_t v = 0;
    bool on = k < MT_N - MT_M;
    if (on) v = mt[k + MT_M] ^ mt_mix(mt[k], mt[k + 1]);
    __syncthreads();
    if (on) mt[k] = v;
    __syncthreads();
  }
  for (int k0 = MT_N - MT_M; k0 < MT_N - 1; k0 += 64) {
    int k = k0 + lane;
    uint32_t v = 0;
    bool on = k < MT_N - 1;
    if (on) v = mt[k + (MT_M - MT_N)] ^ mt_mix(mt[k], mt[k + 1]);
    __syncthreads();
    if (on) mt[k] = v;
    __syncthreads();
  }
  if (lane == 0) mt[MT_N - 1] = mt[MT_M - 1] ^ mt_mix(mt[MT_N - 1], mt[0]);
  __syncthreads();
}

// Regenerate tempered block `which` of game g from the raw state (advancing it one twist).
__device__ inline void wave_refill(const DevBuffers& b, int g, int which, MSB_AS_LDS uint32_t* tmp, int lane) {
  uint32_t* mt = b.rng_mt + (size_t)g * MT_N;
  for (int k = lane; k < MT_N; k += 64) tmp[k] = mt[k];
  __syncthreads();
  wave_twist_lds(tmp, lane);
  uint32_t* out = b.rng_out + (size_t)g * RNG_WORDS + which * MT_N;
  for (int k = lane; k < MT_N; k += 64) {
    uint32_t v = tmp[k];
    mt[k] = v;
    out[k] = mt_temper(v);
  }
  __syncthreads();
}

// Attach game g's stream window to the record an engine works on (fields H_RNGCUR/NXT/POS).
template <class E>
__device__ MSB_INL void attach_rng(E& e, const DevBuffers& b, int g, uint32_t rng) {
  const uint32_t* base = b.rng_out + (size_t)g * RNG_WORDS;
  int cur = (rng >> 16) & 1;
  e.rng_attach(base + cur * MT_N, base + (cur ^ 1) * MT_N, rng & 0xffffu);
}
__device__ MSB_INL uint32_t peek_u32(const DevBuffers& b, int g, uint32_t rng) {
  const uint32_t* base = b.rng_out + (size_t)g * RNG_WORDS;
  int cur = (rng >> 16) & 1;
  uint32_t pos = rng & 0xffffu;
  return pos < (uint32_t)MT_N ? base[cur * MT_N + pos] : base[(cur ^ 1) * MT_N + pos - MT_N];
}

// Serial form for the lane-per-game API kernels: one lane owns the game.
__device__ inline void lane_commit_rng(const DevBuffers& b, int g, GameMeta& m, uint32_t pos) {
  int cur = (m.rng >> 16) & 1;
  if (pos >= (uint32_t)MT_N) {
    pos -= MT_N;
    uint32_t* mt = b.rng_mt + (size_t)g * MT_N;
    mt_twist(mt);
    uint32_t* out = b.rng_out + (size_t)g * RNG_WORDS + cur * MT_N;
    for (int k = 0; k < MT_N; k++) out[k] = mt_temper(mt[k]);
    cur ^= 1;
  }
  m.rng = pos | ((uint32_t)cur << 16);
}

// [16,928) holds the function-scope counters of the profiling build (msb_base.h); then the weight table (state.h).
constexpr int LDS_ORIGIN = LDS_RECORDS;

// ------------------------------------------------------------------------------------------------
// Hot kernel: a wavefront takes a game, keeps its record in LDS and plays up to `rounds` decisions of it (look-ahead +
// score + argmax + commit each) before it writes the record back and takes the next game.
// Dynamic LDS map (bytes): weight table | [PRIV, +SG*U*16) candidate records, lane-interleaved in 16-byte granules |
// the game's current record | 10 weights + 10 "before" + 10 "best after" features |
// the candidates' "after" features | the candidates' work stacks (SKW words each, interleaved word by word)
// ------------------------------------------------------------------------------------------------
__device__ MSB_INL int nth_set_bit(const uint64_t mask[3], int k) {
  for (int w = 0; w < 3; w++) {
    int c = __popcll(mask[w]);
    if (k < c) {
      uint64_t m = mask[w];
      for (int i = 0; i < k; i++) m &= m - 1;
      return w * 64 + __ffsll((long long)m) - 1;
    }
    k -= c;
  }
  return -1;
}

template <int U>
struct DecideLds {
  static constexpr int PRIV = LDS_ORIGIN;
  static constexpr int PRIV_BYTES = SG * U * 16 > MT_N * 4 ? SG * U * 16 : ((MT_N * 4 + 15) & ~15);   // doubles as the twist buffer
  static constexpr int PAR = PRIV + PRIV_BYTES;      // the game's current record
  static constexpr int WF = PAR + SG * 16;           // 10 weights + 10 "before" features + 10 features of the best successor (f64)
  static constexpr int SKB = WF + 240;               // work stacks of the U candidate lanes ...
  static constexpr int CF = SKB;                     // ... and, once a pass has stepped (stacks empty), their ten "after" features each (f64)
  static constexpr int TOTAL = SKB + U * SKW * 4;
};

__device__ MSB_INL unsigned long long uni64(unsigned long long v) {   // a wave-uniform 64-bit value into scalar registers
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Up to `rounds` decisions of game g by the calling wavefront.  The record lives in LDS from the first decision to
// the last; HBM sees one read and one write of it per call.
template <int U>
__device__ MSB_INL void play_game(const DevBuffers& b, const int g, const int lane, int max_turns, int rounds, int write_scores) {
  typedef DecideLds<U> L;
  typedef Engine<SharedMem<L::PAR>> ParEngine;
  typedef Engine<LaneMem<U, L::PRIV, L::SKB, SKW>> CandEngine;
  GameMeta meta = b.meta[g];
  if (meta.result != -2) {
    if (lane == 0) {
      b.meta[g].last_action = 255;
      if (b.best) b.best[g] = NAN;
    }
    return;
  }
  PROF_DECL();
  MSB_AS_LDS u32x4* par = (MSB_AS_LDS u32x4*)(uintptr_t)L::PAR;
  MSB_AS_LDS u32x4* priv = (MSB_AS_LDS u32x4*)(uintptr_t)L::PRIV;
  MSB_AS_LDS double* wf = (MSB_AS_LDS double*)(uintptr_t)L::WF;
  MSB_AS_LDS double* cf = (MSB_AS_LDS double*)(uintptr_t)(L::CF + (lane < U ? lane : 0) * 80);   // this candidate lane's features
  u32x4* grec = (u32x4*)(b.state + (size_t)g * SW);
  for (int c = lane; c < SG; c += 64) par[c] = grec[c];   // one coalesced 16-B-per-lane pass
  __syncthreads();
  ParEngine pe;
  CandEngine ce;
  if (lane == 0) attach_rng(pe, b, g, meta.rng);
  __syncthreads();
  bool have_before = false;   // wf[10..19] holds the features of the CURRENT state (the committed successor's)
  double last_score = NAN;
  int played = 0;
  PROF_MARK(0);   // stage

  for (int round = 0; round < rounds; round++) {
    // rollout contract: while not have_winner() and steps < max_turns
    if (pe.have_winner() || meta.steps >= max_turns) {
      int b0 = pe.pl_base(0), b1 = pe.pl_base(1);
      int res = -1;
      if (pe.have_winner()) res = (b1 < 0 && b0 >= 0) ? 0 : (b0 < 0 && b1 >= 0) ? 1 : -1;
      meta.result = (int8_t)res;
      if (pe.have_winner()) meta.flags |= 1;
      if (played == 0) {
        meta.last_action = 255;
        last_score = NAN;
      }
      break;
    }
#if defined(MSB_STUDY_LEGAL)
    {   // study build (scripts/step_cost.sh): the legal mask computed once more
      msb_u64x4 lm2 = pe.legal_mask_v();
      asm volatile("" : : "v"(lm2[0]), "v"(lm2[1]), "v"(lm2[2]) : "memory");
    }
#endif
    const msb_u64x4 lm = pe.legal_mask_v();
    // the legal set as wave-uniform scalars; `rem` loses the U lowest actions after every pass, so a lane finds its
    // action among the first U set bits (at most U - 1 steps, on the scalar unit for the common part)
    const uint64_t mask[3] = {uni64(lm[0]), uni64(lm[1]), uni64(lm[2])};
    uint64_t rem[3] = {mask[0], mask[1], mask[2]};
    const int n_legal = __popcll(mask[0]) + __popcll(mask[1]) + __popcll(mask[2]);
    PROF_MARK(1);   // legal mask
    const bool before_raises = pe.observation_raises();
    // weights and "before" features are parked in LDS: 40 fewer live VGPRs across the recursive step calls
    {
      const double* wt = b.weights + (size_t)(pe.local() == 0 ? meta.p1 : meta.p2) * 10;
      if (lane < 10) wf[lane] = wt[lane];
      // The "before" features of this decision are the "after" features the previous decision computed for the
      // successor it committed (same state, same mover); only the first decision of a call computes them.
      if (!before_raises && !have_before) {
        if (lane == 0) pe.features(wf + 10);
      }
    }
    __syncthreads();
    PROF_MARK(2);   // before-features

    // Running best over the passes (uniform across the wave).  When the legal set needs more than
    // one pass, the best successor so far is parked in the game's record in HBM so that nothing is replayed.
    // (Parking it in its own column and running later passes on the other U - 1 columns saves that record and the
    // copy, but measured 6 % slower: more passes.)
    constexpr int NONE_A = 1 << 20;
    double run_s = 0.0;
    int run_a = NONE_A;
    int cfault = 0;
    int feat_ok = 0;                  // wf[20..29] holds the features of the best successor so far
    int la_fault = 0;                 // first build-limit fault a look-ahead of this decision hit
    int wl = 0;                       // column (lane) holding the committed successor
    const bool multi = n_legal > U;
    for (int base = 0; base < n_legal; base += U) {
      int k = base + lane;
      double s = 0.0;   // except Exception -> 0.0 (evo/heuristic_agent.py:48-51)
      int a = NONE_A;
      int my_fault = 0;
      int my_feat = 0;
      int f = 0;
      bool raises = false;
      // copy.deepcopy (stream window included) for the whole pass, by all 64 lanes: granule idx of the interleaved
      // candidate image is record granule idx / U for column idx % U
      {
        const int n_act = n_legal - base < U ? n_legal - base : U;
        __syncthreads();
        for (int idx = lane; idx < SG * U; idx += 64)
          if ((idx & (U - 1)) < n_act) priv[idx] = par[idx / U];
        __syncthreads();
      }
      const bool active = lane < U && k < n_legal;
      if (active) a = nth_set_bit(rem, lane);
      for (int i = 0; i < U; i++) {   // uniform: drop this pass's actions
        if (rem[0]) rem[0] &= rem[0] - 1;
        else if (rem[1]) rem[1] &= rem[1] - 1;
        else rem[2] &= rem[2] - 1;
      }
#if defined(MSB_STUDY_REPEAT)
      // study build only (scripts/step_cost.sh): the look-ahead step (or a prefix of it, MSB_STUDY_CUT_AT) and its clone
      // executed once more, so that the difference of two counter runs is the cost of exactly that
      for (int rep = 0; rep < MSB_STUDY_REPEAT; rep++) {
#if defined(MSB_STUDY_CUT_AT)
        if (active) ce.step(a, MSB_STUDY_CUT_AT);
#else
        if (active) ce.step(a);
#endif
        const int n_act = n_legal - base < U ? n_legal - base : U;
        __syncthreads();
        for (int idx = lane; idx < SG * U; idx += 64)
          if ((idx & (U - 1)) < n_act) priv[idx] = par[idx / U];
        __syncthreads();
      }
#endif
      PROF_MARK(3);   // clone
      if (active) {
        ce.step(a);
        f = ce.fault();
        raises = f == 0 && ce.observation_raises();
      }
      PROF_MARK(4);   // step
      if (active) {
        if (f == 0 && !before_raises && !raises) {
          // the ten values go to LDS for the winner's sake (argmax below) and feed the score from registers; they are
          // dead before the shuffles, so they never sit in registers across them
          double fa[10];
#if defined(MSB_STUDY_FEATURES)
          {   // study build: the candidate's features and its score computed once more
            double fb[10];
            ce.features(fb);
            double s2 = CandEngine::action_score_lds(wf, fb);
            asm volatile("" : : "v"(s2), "v"(fb[0]), "v"(fb[9]) : "memory");
          }
#endif
          ce.features(fa);
          for (int i = 0; i < 10; i++) cf[i] = fa[i];
          s = CandEngine::action_score_lds(wf, fa);
          my_feat = 1;
        }
        if (write_scores) b.scores[(size_t)g * MONSOON_NUM_ACTIONS + a] = s;
        my_fault = f ? f : (raises ? FAULT_INT_CARD : 0);
      }
      PROF_MARK(5);   // after-features + score
      {
        // A look-ahead that hits a limit of this build scores 0.0 where the reference would compute a score: the game
        // is marked (meta.la_fault), counted (monsoon_stats.lookahead_capacity_faults), reported (monsoon_game_faults).
        const unsigned long long lfb = __ballot(active && f >= FAULT_CAPACITY);
        if (lfb && !la_fault) la_fault = __builtin_amdgcn_readlane(f, __builtin_ctzll(lfb));
      }
      // first maximum over the ascending legal list == (max score, then min action id)
      // only lanes 0..U-1 hold candidates: butterfly over those, then broadcast lane 0's result to the wave
      double cs = s;
      int ca = a;
      for (int off = U / 2; off >= 1; off >>= 1) {
        double os = __shfl_xor(cs, off);
        int oa = __shfl_xor(ca, off);
        bool take = (oa != NONE_A) && (ca == NONE_A || os > cs || (os == cs && oa < ca));
        if (take) {
          cs = os;
          ca = oa;
        }
      }
      cs = __longlong_as_double((long long)uni64((unsigned long long)__double_as_longlong(cs)));
      ca = __builtin_amdgcn_readfirstlane(ca);
      if (ca != NONE_A && (run_a == NONE_A || cs > run_s)) {   // later passes hold larger action ids: strict >
        run_s = cs;
        run_a = ca;
        unsigned long long bal = __ballot(a == ca);
        wl = __ffsll((long long)bal) - 1;
        cfault = __builtin_amdgcn_readlane(my_fault, wl);
        feat_ok = __builtin_amdgcn_readlane(my_feat, wl);
        if (feat_ok && lane < 10)   // the winner keeps its features: the next decision's "before" side
          wf[20 + lane] = ((MSB_AS_LDS const double*)(uintptr_t)L::CF)[wl * 10 + lane];
        if (multi) {
          // park it in the game's record in HBM: that slot is free while the live record sits in LDS (one coalesced
          // 16-byte-per-lane store; a spare LDS record here would cost every wavefront a record's worth of LDS = two wavefronts per CU)
          __syncthreads();
          for (int c = lane; c < SG; c += 64) grec[c] = priv[c * U + wl];
        }
      }
      PROF_MARK(6);   // argmax + park
    }
    __syncthreads();
    // commit: adapter = adapter.apply_action(best).  The successor carries its own stream cursor (H_RNGPOS).
    if (multi) {
      // read back what this wavefront parked: agent-scope loads, so that the per-CU vector cache cannot answer with an
      // older line of the record
      for (int c = lane; c < 2 * SG; c += 64) {
        unsigned long long v = __hip_atomic_load((unsigned long long*)grec + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ((MSB_AS_LDS unsigned long long*)par)[c] = v;
      }
    } else {
      for (int c = lane; c < SG; c += 64) par[c] = priv[c * U + wl];
    }
    __syncthreads();
    {
      uint32_t new_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pe.rng_pos());
      int cur = (meta.rng >> 16) & 1;
      if (new_pos >= (uint32_t)MT_N) {
        new_pos -= MT_N;
        wave_refill(b, g, cur, (MSB_AS_LDS uint32_t*)priv, lane);   // the used-up block becomes the new "next" block
        cur ^= 1;
        if (lane == 0) pe.rng_block_advance();
      }
      meta.rng = new_pos | ((uint32_t)cur << 16);
      if (lane == 0) attach_rng(pe, b, g, meta.rng);
      // the committed successor's features become the "before" side of the next decision
      have_before = feat_ok != 0;
      if (have_before && lane < 10) wf[10 + lane] = wf[20 + lane];
      __syncthreads();
    }
    meta.steps++;
    meta.last_action = (uint8_t)run_a;
    meta.lookahead += (uint32_t)n_legal;   // every legal action is stepped exactly once; the commit re-executes nothing
    meta.decided++;   // statistics are per-game fields reduced on demand (k_stats): no same-address atomics here
    if (la_fault && !meta.la_fault) meta.la_fault = (uint8_t)la_fault;
    last_score = run_s;
    played++;
    PROF_MARK(7);   // commit + refill
    if (cfault) {
      // evo/fitness.py:208-210: an exception while applying the action ends the game as a draw
      meta.fault = (uint8_t)cfault;
      meta.result = -1;
      break;
    }
  }
  __syncthreads();
  for (int c = lane; c < SG; c += 64) grec[c] = par[c];
  if (lane == 0) {
    b.meta[g] = meta;
    if (b.best) b.best[g] = last_score;
  }
  PROF_FLUSH();
}

// Persistent wavefronts: the grid is what the GPU holds at once.  The games are split into POP_PARTS contiguous
// ranges; wavefront w works on range w % POP_PARTS (workgroups are dealt to the 8 XCDs round-robin, so a range stays
// on one XCD and its L2): it starts with the game given by its index and then pops further ones from the range's
// counter, the pop being issued before the current game is played so that its latency is hidden.  Games stay in
// index order -- neighbouring records, stream blocks and meta rows are touched together; sorting the games by
// expected cost was measured 5-8 % slower.  One counter per range, 128 bytes apart: atomics on ONE address serialise
// at ~25 ns each, which capped a launch at 65 536 x 25 ns (the same trap as per-game statistics counters; see
// k_stats).  Every wave reaches its exit (t >= hi): counters only grow.  b.pop[parity] is this launch's set; the
// other one is cleared for the next launch.  persistent = 0: one workgroup per game.  The host launches the
// persistent form only with at least POP_PARTS workgroups (a range without a wavefront would never be played).
// rounds = 1 is one decision round over the batch; rounds > max_turns plays every game to its end (rollouts).
constexpr int POP_PARTS = 8, POP_STRIDE = 32;
template <int U, int WPE>
__global__ void __launch_bounds__(64, WPE) k_play(DevBuffers b, int n, int max_turns, int rounds, int write_scores, int persistent, int parity) {
  const int lane = threadIdx.x;
  lds_init_wtab(b.wk_ovf + (size_t)blockIdx.x * (U * OVF_WORDS));
  // one body for both forms (play_game is the whole rules core, inlined once): the non-persistent form is a "range" of
  // one game that is never refilled
  int* mine = b.pop + parity * POP_PARTS * POP_STRIDE;
  int* other = b.pop + (parity ^ 1) * POP_PARTS * POP_STRIDE;
  if (persistent && blockIdx.x == 0 && lane < POP_PARTS) other[lane * POP_STRIDE] = 0;
  const int part = blockIdx.x % POP_PARTS, rank = blockIdx.x / POP_PARTS;
  const int waves = ((int)gridDim.x - part + POP_PARTS - 1) / POP_PARTS;   // wavefronts working on this range
  const int lo = persistent ? (int)((long long)n * part / POP_PARTS) : 0;
  const int hi = persistent ? (int)((long long)n * (part + 1) / POP_PARTS) : n;
  int t = persistent ? lo + rank : (int)blockIdx.x;
  while (t < hi) {
    int nxt = 0x7fffffff;
    if (persistent && lane == 0) nxt = lo + waves + atomicAdd(&mine[part * POP_STRIDE], 1);
    play_game<U>(b, t, lane, max_turns, rounds, write_scores);
    __syncthreads();   // the LDS image is reused by the next game
    t = __builtin_amdgcn_readfirstlane(nxt);
  }
}

// What the host needs to launch one variant (variant.hip defines one getter per instantiation).
struct VariantOps {
  int lanes, wpe;           // U, W
  int kind;                 // variants.def's third column: 1 k_play, 2 / 4 k_play_multi, 10 k_play_reg
  int games;                // games a wavefront plays at once (k_play_multi: 2 or 4; else 1)
  int lds_bytes;            // dynamic LDS of one workgroup
  hipError_t (*occupancy)(int* blocks_per_cu, int lds_bytes);
  void (*play)(int grid, int lds_bytes, hipStream_t stream, DevBuffers b, int n, int max_turns, int rounds, int write_scores, int persistent,
               int parity);
};

}  // namespace msbk
