// libmonsoon_hip.so -- MI355X (gfx950) batched Stormbound engine: kernels + C ABI (include/monsoon.h).
//
// Execution model
//   * Hot kernel k_decide<U>: ONE WAVEFRONT PER GAME (a persistent grid of resident wavefronts, each popping game
//     indices from its range's counter).  The game's record (992 B) is staged from HBM into LDS with one coalesced
//     pass; the legal-action mask is evaluated on that shared copy (LDS broadcast reads), the "before" features
//     are the ones the previous decision computed for the successor it committed; then up to U candidate actions
//     are advanced at once, lane l stepping its own private copy of the state.  The private copies are interleaved
//     across lanes in 16-byte granules (granule c of lane l at (c*U + l)*16), so lanes touching the same field hit
//     distinct LDS banks and a whole entity is one ds_read_b128.  Scores are reduced with shuffles over the U
//     candidate lanes (first maximum in ascending action order = np.argmax over the sorted legal list) and the
//     winner's column is written back as the game's new record.  Nothing is re-executed: the committed successor
//     IS one of the look-ahead results (when the legal set needs several passes of U lanes, the best successor so
//     far is parked in a spare LDS column).
//   * The game's MT19937 stream lives in HBM as two blocks of tempered outputs (current + next)
//     plus the raw state; candidate steps read it through a private cursor, the committed
//     cursor is stored back and the wave regenerates a block (twist in LDS) when it is used up.
//   * API kernels (reset/step/legal/observe/features/status/export) map one LANE per game with the
//     same LDS layout (U = 64); they back the batch=1 Game view and the parity tests.
//   * Integer/index work: no MFMA.  f64 appears only in the weighted draw and the score.
//
// There is no CPU path in this library.  A missing/unsupported device is an error.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/monsoon.h"
#include "canon.h"

using namespace msb;

namespace {

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
                           // b1: cfeat[last_action] holds the features of the CURRENT state (it was the committed successor)
  uint16_t steps;          // committed steps (decisions and monsoon_step calls)
  uint16_t decided;        // decisions committed by k_decide
  uint32_t rng;            // cursor (bits 0-15) | current block (bit 16)
  uint32_t lookahead;      // look-ahead transitions executed for this game
  uint32_t match;          // schedule index (rollout)
};

struct DevBuffers {
  uint32_t* state;     // [cap][SW]
  uint32_t* rng_out;   // [cap][2][624]
  uint32_t* rng_mt;    // [cap][624]
  GameMeta* meta;      // [cap]
  double* weights;     // [n_individuals][10]
  unsigned long long* stats;  // [8]: lookahead, decisions, finished, faults, capacity_faults
  double* scores;      // [cap][156] or null
  double* best;        // [cap]
  double* cfeat;       // [cap][156][10] features of every look-ahead successor of the last decision, by action id
  int* pop;            // [2][POP_PARTS * POP_STRIDE] game-index counters of the persistent k_decide, alternating between launches
  unsigned long long* prof;   // [cap][..] phase cycles, scope cycles, scope calls (profiling build), profiling build only (else null)
};

enum { ST_LOOKAHEAD = 0, ST_DECISIONS = 1, ST_FINISHED = 2, ST_FAULTS = 3, ST_CAPFAULTS = 4, ST_PROF = 8, ST_WORDS = 32, PROF_WORDS = 138 };
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
__device__ void wave_twist_lds(MSB_AS_LDS uint32_t* mt, int lane) {
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
__device__ void wave_refill(const DevBuffers& b, int g, int which, MSB_AS_LDS uint32_t* tmp, int lane) {
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
__device__ void lane_commit_rng(const DevBuffers& b, int g, GameMeta& m, uint32_t pos) {
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

// ------------------------------------------------------------------------------------------------
// Lane-per-game API kernels.  Block = 64 threads; the records are staged in DYNAMIC LDS starting at LDS
// address 0 (these kernels declare no static __shared__), interleaved across lanes in 16-byte granules.
// ------------------------------------------------------------------------------------------------
// games per 64-thread block in the API kernels: the extended record is too large for 64 LDS columns
#if defined(MSB_EXT) && MSB_EXT
constexpr int API_LANES = 16;
#else
constexpr int API_LANES = 64;
#endif
// LDS address 0 is avoided on purpose: an integer constant 0 cast to an LDS pointer is the null pointer,
// which is not address 0 on this target; every region starts at LDS_ORIGIN.
#if defined(MSB_PROF) && MSB_PROF
constexpr int LDS_ORIGIN = 928;   // [16,928): function-scope counters of the profiling build (msb_base.h)
#else
constexpr int LDS_ORIGIN = 16;
#endif
constexpr int API_LDS_BYTES = LDS_ORIGIN + SG * API_LANES * 16;
typedef LaneMem<API_LANES, LDS_ORIGIN> ApiMem;
typedef Engine<ApiMem> ApiEngine;
#define API_GAME_INDEX()                                  \
  if ((int)threadIdx.x >= API_LANES) return;              \
  int g = blockIdx.x * API_LANES + threadIdx.x;           \
  if (g >= n) return;

__device__ MSB_INL void api_load(const uint32_t* src) {
  const u32x4* s4 = (const u32x4*)src;
  for (int c = 0; c < SG; c++) *(MSB_AS_LDS u32x4*)ApiMem::b(c * 16) = s4[c];
}
__device__ MSB_INL void api_store(uint32_t* dst) {
  u32x4* d4 = (u32x4*)dst;
  for (int c = 0; c < SG; c++) d4[c] = *(MSB_AS_LDS const u32x4*)ApiMem::b(c * 16);
}

__global__ void __launch_bounds__(64) k_seed(DevBuffers b, int n, const uint32_t* seeds) {
  // one wavefront per game: init_genrand is a serial recurrence (lane 0), the two twists are
  // wave-cooperative
  __shared__ uint32_t tmp[MT_N];
  int g = blockIdx.x, lane = threadIdx.x;
  if (g >= n) return;
  MSB_AS_LDS uint32_t* t = (MSB_AS_LDS uint32_t*)tmp;
  if (lane == 0) {
    uint32_t x = seeds[g];
    t[0] = x;
    for (int i = 1; i < MT_N; i++) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      t[i] = x;
    }
  }
  __syncthreads();
  uint32_t* mt = b.rng_mt + (size_t)g * MT_N;
  for (int k = lane; k < MT_N; k += 64) mt[k] = t[k];
  __syncthreads();
  wave_refill(b, g, 0, t, lane);
  wave_refill(b, g, 1, t, lane);
}

__global__ void __launch_bounds__(64) k_init(DevBuffers b, int n, const uint8_t* decks, const uint8_t* factions) {
  API_GAME_INDEX();
  ApiEngine e;
  GameMeta m = b.meta[g];
  m.rng = 0;
  attach_rng(e, b, g, m.rng);
  uint8_t d0[12], d1[12];
  for (int i = 0; i < 12; i++) {
    d0[i] = decks[(size_t)g * 24 + i];
    d1[i] = decks[(size_t)g * 24 + 12 + i];
  }
  e.init_game(d0, d1, factions[2 * g], factions[2 * g + 1]);
  lane_commit_rng(b, g, m, e.rng_pos());
  m.result = -2;
  m.fault = (uint8_t)e.fault();
  m.last_action = 255;
  m.steps = 0;
  m.lookahead = 0;
  b.meta[g] = m;
  api_store(b.state + (size_t)g * SW);
}

__global__ void __launch_bounds__(64) k_legal(DevBuffers b, int n, uint64_t* out) {
  API_GAME_INDEX();
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  msb_u64x4 mask = e.legal_mask_v();
  out[3 * g] = mask[0];
  out[3 * g + 1] = mask[1];
  out[3 * g + 2] = mask[2];
}

__global__ void __launch_bounds__(64) k_step(DevBuffers b, int n, const uint8_t* actions, int8_t* reward, uint8_t* done,
                                              uint8_t* fault, uint8_t* illegal) {
  API_GAME_INDEX();
  int a = actions[g];
  reward[g] = 0;
  done[g] = 0;
  fault[g] = 0;
  illegal[g] = 0;
  if (a == 255) return;
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  msb_u64x4 mask = e.legal_mask_v();
  uint64_t word = a < 64 ? mask[0] : (a < 128 ? mask[1] : mask[2]);
  // PASS (155) is always accepted: the reference's step executes it whenever asked, and its own scripted bot
  // ends a turn with PASS while plays remain (games/stormbound.py:637)
  if (a >= 156 || (a != 155 && !((word >> (a & 63)) & 1))) {
    illegal[g] = 1;
    return;
  }
  GameMeta m = b.meta[g];
  attach_rng(e, b, g, m.rng);
  int rd = e.step(a);
  reward[g] = (int8_t)(rd & 1);
  done[g] = (uint8_t)((rd >> 1) & 1);
  fault[g] = (uint8_t)e.fault();
  lane_commit_rng(b, g, m, e.rng_pos());
  m.steps++;
  m.last_action = (uint8_t)a;
  m.flags &= ~2;   // no cached features for a state reached through monsoon_step
  if (e.fault()) m.fault = (uint8_t)e.fault();
  b.meta[g] = m;
  api_store(b.state + (size_t)g * SW);
}

// Stormbound.expert_action for every game (draws from the game's stream, so the cursor is committed)
__global__ void __launch_bounds__(64) k_expert(DevBuffers b, int n, uint8_t* out_action, uint8_t* fault) {
  API_GAME_INDEX();
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  GameMeta m = b.meta[g];
  attach_rng(e, b, g, m.rng);
  int a = e.expert_action();
  out_action[g] = (uint8_t)a;
  fault[g] = (uint8_t)e.fault();
  lane_commit_rng(b, g, m, e.rng_pos());
  b.meta[g] = m;
}

__global__ void __launch_bounds__(64) k_observe(DevBuffers b, int n, int32_t* out, uint8_t* raises) {
  API_GAME_INDEX();
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  bool r = e.observation_raises();
  raises[g] = r ? 1 : 0;
  if (!r) e.observe(out + (size_t)g * MONSOON_OBS_INTS);
}

__global__ void __launch_bounds__(64) k_features(DevBuffers b, int n, double* out) {
  API_GAME_INDEX();
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  double f[10];
  if (e.observation_raises()) {
    for (int i = 0; i < 10; i++) f[i] = NAN;
  } else {
    e.features(f);
  }
  for (int i = 0; i < 10; i++) out[(size_t)g * 10 + i] = f[i];
}

__global__ void __launch_bounds__(64) k_status(DevBuffers b, int n, int32_t* out) {
  int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= n) return;
  FlatMem fm{(uint8_t*)(b.state + (size_t)g * SW)};
  Engine<FlatMem> e;
  e.m = fm;
  out[4 * g] = e.local();
  out[4 * g + 1] = e.have_winner() ? 1 : 0;
  out[4 * g + 2] = e.pl_base(0);
  out[4 * g + 3] = e.pl_base(1);
}

__global__ void __launch_bounds__(64) k_export(DevBuffers b, int g, uint8_t* out, int32_t* len) {
  if (threadIdx.x != 0) return;
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  *len = canon_record(e, peek_u32(b, g, b.meta[g].rng), out);
}

__global__ void __launch_bounds__(64) k_hash(DevBuffers b, int n, uint64_t* out) {
  API_GAME_INDEX();
  ApiEngine e;
  api_load(b.state + (size_t)g * SW);
  uint8_t rec[CANON_MAX];
  int len = canon_record(e, peek_u32(b, g, b.meta[g].rng), rec);
  out[g] = fnv1a64(rec, len);
}

// ------------------------------------------------------------------------------------------------
// Hot kernel: one decision (look-ahead + score + argmax + commit) per game, one wavefront per game.
// Dynamic LDS map (bytes):  [0, SG*U*16) candidate records, lane-interleaved | parent record | best
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
  static constexpr int PAR = PRIV + PRIV_BYTES;
  static constexpr int BEST = PAR + SG * 16;
  static constexpr int WF = BEST + SG * 16;          // 10 weights + 10 "before" features (f64), shared by the lanes
  static constexpr int TOTAL = WF + 160;
};

// One decision of game g by the calling wavefront.
template <int U>
__device__ void decide_game(const DevBuffers& b, const int g, const int lane, int max_turns, int write_scores) {
  typedef DecideLds<U> L;
  typedef Engine<SharedMem<L::PAR>> ParEngine;
  typedef Engine<LaneMem<U, L::PRIV>> CandEngine;
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
  MSB_AS_LDS u32x4* bestcol = (MSB_AS_LDS u32x4*)(uintptr_t)L::BEST;
  u32x4* grec = (u32x4*)(b.state + (size_t)g * SW);
  for (int c = lane; c < SG; c += 64) par[c] = grec[c];   // one coalesced 16-B-per-lane pass
  __syncthreads();

  ParEngine pe;
  if (lane == 0) attach_rng(pe, b, g, meta.rng);
  __syncthreads();

  // rollout contract: while not have_winner() and steps < max_turns
  if (pe.have_winner() || meta.steps >= max_turns) {
    if (lane == 0) {
      int b0 = pe.pl_base(0), b1 = pe.pl_base(1);
      int res = -1;
      if (pe.have_winner()) res = (b1 < 0 && b0 >= 0) ? 0 : (b0 < 0 && b1 >= 0) ? 1 : -1;
      meta.result = (int8_t)res;
      meta.last_action = 255;
      if (pe.have_winner()) meta.flags |= 1;
      b.meta[g] = meta;
      if (b.best) b.best[g] = NAN;
    }
    return;
  }

  PROF_MARK(0);   // stage
  const msb_u64x4 lm = pe.legal_mask_v();
  // the legal set as wave-uniform scalars; `rem` loses the U lowest actions after every pass, so a lane finds its
  // action among the first U set bits (at most U - 1 steps, on the scalar unit for the common part)
  auto uni64 = [](unsigned long long v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  const uint64_t mask[3] = {uni64(lm[0]), uni64(lm[1]), uni64(lm[2])};
  uint64_t rem[3] = {mask[0], mask[1], mask[2]};
  const int n_legal = __popcll(mask[0]) + __popcll(mask[1]) + __popcll(mask[2]);
  PROF_MARK(1);   // legal mask
  const bool before_raises = pe.observation_raises();
  // weights and "before" features are parked in LDS: 40 fewer live VGPRs across the recursive step calls
  MSB_AS_LDS double* wf = (MSB_AS_LDS double*)(uintptr_t)L::WF;
  {
    const double* wt = b.weights + (size_t)(pe.local() == 0 ? meta.p1 : meta.p2) * 10;
    if (lane < 10) wf[lane] = wt[lane];
    // The "before" features of this decision are the "after" features the previous decision computed for the
    // successor it committed (same state, same mover): they were kept by action id.
    if (!before_raises) {
      if ((meta.flags & 2) && meta.last_action < MONSOON_NUM_ACTIONS) {
        if (lane < 10) wf[10 + lane] = b.cfeat[((size_t)g * MONSOON_NUM_ACTIONS + meta.last_action) * 10 + lane];
      } else {
        double fb[10];
        pe.features(fb);
        if (lane == 0)
          for (int i = 0; i < 10; i++) wf[10 + i] = fb[i];
      }
    }
  }
  __syncthreads();
  PROF_MARK(2);   // before-features

  CandEngine ce;
  // Running best over the passes (uniform across the wave).  When the legal set needs more than
  // one pass, the best successor so far is parked in a spare LDS column so that nothing is replayed.
  constexpr int NONE_A = 1 << 20;
  double run_s = 0.0;
  int run_a = NONE_A;
  uint32_t new_pos = 0;
  int cfault = 0;
  int feat_ok = 0;                  // the committed successor's features are in cfeat[A]
  int wl = 0;                       // column (lane) holding the committed successor
  const bool multi = n_legal > U;
  for (int base = 0; base < n_legal; base += U) {
    int k = base + lane;
    double s = 0.0;   // except Exception -> 0.0 (evo/heuristic_agent.py:48-51)
    int a = NONE_A;
    uint32_t my_pos = 0;
    int my_fault = 0;
    int my_feat = 0;
    int f = 0;
    bool raises = false;
    // copy.deepcopy (stream window included) for the whole pass, by all 64 lanes: granule idx of the interleaved
    // candidate image is parent granule idx / U for column idx % U
    {
      const int n_act = n_legal - base < U ? n_legal - base : U;
      __syncthreads();
      for (int idx = lane; idx < SG * U; idx += 64)
        if ((idx & (U - 1)) < n_act) priv[idx] = par[idx / U];
      __syncthreads();
    }
    if (lane < U && k < n_legal) a = nth_set_bit(rem, lane);
    for (int i = 0; i < U; i++) {   // uniform: drop this pass's actions
      if (rem[0]) rem[0] &= rem[0] - 1;
      else if (rem[1]) rem[1] &= rem[1] - 1;
      else rem[2] &= rem[2] - 1;
    }
    PROF_MARK(3);   // clone
    if (lane < U && k < n_legal) {
      ce.step(a);
      f = ce.fault();
      raises = f == 0 && ce.observation_raises();
    }
    PROF_MARK(4);   // step
    if (lane < U && k < n_legal) {
      if (f == 0 && !before_raises && !raises) {
        double fa[10], wv[10], fbv[10];
        ce.features(fa);
        for (int i = 0; i < 10; i++) {
          wv[i] = wf[i];
          fbv[i] = wf[10 + i];
        }
        s = CandEngine::action_score(wv, fbv, fa);
        double* slot = b.cfeat + ((size_t)g * MONSOON_NUM_ACTIONS + a) * 10;
        for (int i = 0; i < 10; i++) slot[i] = fa[i];
        my_feat = 1;
      }
      if (write_scores) b.scores[(size_t)g * MONSOON_NUM_ACTIONS + a] = s;
      my_pos = ce.rng_pos();
      my_fault = f ? f : (raises ? FAULT_INT_CARD : 0);
    }
    PROF_MARK(5);   // after-features + score
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
    {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(cs);
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bits);
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bits >> 32));
      cs = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      ca = __builtin_amdgcn_readfirstlane(ca);
    }
    if (ca != NONE_A && (run_a == NONE_A || cs > run_s)) {   // later passes hold larger action ids: strict >
      run_s = cs;
      run_a = ca;
      unsigned long long bal = __ballot(a == ca);
      wl = __ffsll((long long)bal) - 1;
      new_pos = (uint32_t)__builtin_amdgcn_readlane((int)my_pos, wl);
      cfault = __builtin_amdgcn_readlane(my_fault, wl);
      feat_ok = __builtin_amdgcn_readlane(my_feat, wl);
      if (multi) {
        __syncthreads();
        for (int c = lane; c < SG; c += 64) bestcol[c] = priv[c * U + wl];
        __syncthreads();
      }
    }
    PROF_MARK(6);   // argmax + park
  }
  const int A = run_a;
  const double rs = run_s;
  __syncthreads();
  // commit: adapter = adapter.apply_action(best)
  if (multi) {
    for (int c = lane; c < SG; c += 64) grec[c] = bestcol[c];
  } else {
    for (int c = lane; c < SG; c += 64) grec[c] = priv[c * U + wl];
  }
  __syncthreads();
  int cur = (meta.rng >> 16) & 1;
  if (new_pos >= (uint32_t)MT_N) {
    new_pos -= MT_N;
    wave_refill(b, g, cur, (MSB_AS_LDS uint32_t*)priv, lane);   // the used-up block becomes the new "next" block
    cur ^= 1;
  }
  if (lane == 0) {
    meta.rng = new_pos | ((uint32_t)cur << 16);
    meta.steps++;
    meta.last_action = (uint8_t)A;
    int executed = n_legal;   // every legal action is stepped exactly once; the commit re-executes nothing
    meta.lookahead += (uint32_t)executed;
    meta.flags = (uint8_t)((meta.flags & ~2) | (feat_ok ? 2 : 0));
    meta.decided++;   // statistics are per-game fields reduced on demand (k_stats): no same-address atomics here
    if (cfault) {
      // evo/fitness.py:208-210: an exception while applying the action ends the game as a draw
      meta.fault = (uint8_t)cfault;
      meta.result = -1;
    }
    b.meta[g] = meta;
    if (b.best) b.best[g] = rs;
  }
  PROF_MARK(7);   // commit + refill
  PROF_FLUSH();
}

// Hot kernel.  Persistent wavefronts: the grid is what the GPU holds at once.  The games are split into
// POP_PARTS contiguous ranges; wavefront w works on range w % POP_PARTS (workgroups are dealt to the 8 XCDs round-
// robin, so a range stays on one XCD and its L2): it starts with the game given by its index and then pops further
// ones from the range's counter, the pop being issued before the current game is played so that its latency is
// hidden.  Games stay in index order -- neighbouring records, stream blocks and meta rows are touched together;
// sorting the games by expected cost was measured 5-8 % slower.  One counter per range, 128 bytes apart: atomics
// on ONE address serialise at ~25 ns each, which capped the whole launch at 65 536 x 25 ns (the same trap as
// per-game statistics counters; see k_stats).  Every wave reaches its exit (t >= hi): counters only grow.
// b.pop[parity] is this launch's set; the other one is cleared for the next launch.  persistent = 0: one workgroup
// per game.
constexpr int POP_PARTS = 8, POP_STRIDE = 32;
template <int U, int WPE>
__global__ void __launch_bounds__(64, WPE) k_decide(DevBuffers b, int n, int max_turns, int write_scores, int persistent, int parity) {
  const int lane = threadIdx.x;
  if (!persistent) {
    if ((int)blockIdx.x < n) decide_game<U>(b, blockIdx.x, lane, max_turns, write_scores);
    return;
  }
  int* mine = b.pop + parity * POP_PARTS * POP_STRIDE;
  int* other = b.pop + (parity ^ 1) * POP_PARTS * POP_STRIDE;
  if (blockIdx.x == 0 && lane < POP_PARTS) other[lane * POP_STRIDE] = 0;
  const int part = blockIdx.x % POP_PARTS, rank = blockIdx.x / POP_PARTS;
  const int waves = ((int)gridDim.x - part + POP_PARTS - 1) / POP_PARTS;   // wavefronts working on this range
  const int lo = (int)((long long)n * part / POP_PARTS), hi = (int)((long long)n * (part + 1) / POP_PARTS);
  int t = lo + rank;
  while (t < hi) {
    int nxt = 0x7fffffff;
    if (lane == 0) nxt = lo + waves + atomicAdd(&mine[part * POP_STRIDE], 1);
    decide_game<U>(b, t, lane, max_turns, write_scores);
    __syncthreads();   // the LDS image is reused by the next game
    t = __builtin_amdgcn_readfirstlane(nxt);
  }
}

// Statistics of the loaded games, reduced from the per-game fields: {look-ahead steps, decisions, games ended by a
// winner, games stopped by a fault, of those build-limit faults}.  One atomic per wavefront into a zeroed buffer.
__global__ void __launch_bounds__(256) k_stats(DevBuffers b, int n, unsigned long long* out) {
  unsigned long long v[5] = {0, 0, 0, 0, 0};
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
    GameMeta m = b.meta[g];
    v[0] += m.lookahead;
    v[1] += m.decided;
    v[2] += (m.flags & 1) ? 1 : 0;
    v[3] += (m.result == -1 && m.fault) ? 1 : 0;
    v[4] += (m.result == -1 && m.fault >= FAULT_CAPACITY) ? 1 : 0;
  }
  for (int i = 0; i < 5; i++) {
    unsigned long long x = v[i];
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out[i], x);
  }
}

__global__ void k_faults(DevBuffers b, int n, uint8_t* out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = b.meta[g].fault;
}

__global__ void k_clear_scores(double* scores, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scores[i] = NAN;
}

__global__ void k_assign(DevBuffers b, int n, const int32_t* p1, const int32_t* p2) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  b.meta[g].p1 = p1[g];
  b.meta[g].p2 = p2[g];
}

// Count live games (rollout loop exit test) and collect per-individual tallies.
__global__ void k_count_live(DevBuffers b, int n, int* live) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  if (b.meta[g].result == -2) atomicAdd(live, 1);
}

__global__ void k_collect(DevBuffers b, int n, int32_t* counts, int8_t* results, int32_t* steps) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = g < n;
  GameMeta m = b.meta[on ? g : 0];
  int r = m.result == -2 ? -1 : m.result;
  // evo/fitness.py:160-166: wins + 0.5*draws for the row individual (p1).  Consecutive games usually share their row
  // individual: the lanes of a wave are grouped by p1 and one lane adds the group's totals (an atomicAdd per game on
  // the same address serialises at ~25 ns each).
  const int lane = threadIdx.x & 63;
  for (unsigned long long todo = __ballot(on); todo;) {
    const int leader = __builtin_ctzll(todo);
    const int key = __builtin_amdgcn_readlane(m.p1, leader);
    const bool mine = on && m.p1 == key;
    const unsigned long long grp = __ballot(mine);
    const int wins = __popcll(__ballot(mine && r == 0)), draws = __popcll(__ballot(mine && r == -1));
    if (lane == leader) {
      if (wins) atomicAdd(&counts[3 * key + 0], wins);
      if (draws) atomicAdd(&counts[3 * key + 1], draws);
      atomicAdd(&counts[3 * key + 2], __popcll(grp));
    }
    todo &= ~grp;
  }
  if (!on) return;
  if (results) results[m.match] = (int8_t)r;
  if (steps) steps[m.match] = m.steps;
}

__global__ void k_set_match(DevBuffers b, int n, int base) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  b.meta[g].match = base + g;
}

}  // namespace

// ================================================================================================
// Host side
// ================================================================================================
struct monsoon {
  monsoon_config cfg;
  int device;
  hipStream_t stream;
  DevBuffers b;
  int wpe;            // k_decide variant: __launch_bounds__ waves per SIMD
  int parity;         // which of b.pop the next k_decide launch uses
  unsigned long long st_acc[5], st_base[5];   // statistics: totals of earlier batches, baseline of the loaded one
  int grid_waves;     // persistent grid size of k_decide (resident wavefronts), 0 = not yet queried
  int n;              // games loaded by the last reset
  int n_individuals;
  std::string err;
  // scratch device buffers for API calls
  uint8_t* d_bytes;   // cap * max(24, 1024/…)
  uint8_t* d_decks;   // [cap][24]
  uint8_t* d_factions;
  uint32_t* d_seeds;
  uint64_t* d_masks;
  int32_t* d_i32;     // cap * 540
  double* d_f64;      // cap * 10
  int32_t* d_p1;
  int32_t* d_p2;
  int* d_int;
  hipEvent_t ev0, ev1;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // decide-kernel timing pairs
  double kernel_ms;
  long long kernel_launches;
};

static std::string g_create_error;

#define HIP_TRY(h, call)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return MONSOON_ERR_DEVICE;                                                          \
    }                                                                                     \
  } while (0)

static const char* kCardIds[NUM_CARDS] = {
#define X(id) #id,
#include "card_id_strings.inc"
#undef X
};

// Cards whose abilities this build does not restate yet (abilities.inc header).
static bool card_unsupported(int c) {
#if defined(MSB_EXT) && MSB_EXT
  (void)c;
  return false;
#else
  return c == C_UA20 || c == C_B005;   // need the extended record: build libmonsoon_hip_ext.so
#endif
}

extern "C" {

int monsoon_version(void) {
#if defined(MSB_EXT) && MSB_EXT
  return 0x10001;   // bit 16: extended record
#else
  return 1;
#endif
}

int monsoon_card_index(const char* id) {
  if (!id) return -1;
  for (int i = 0; i < NUM_CARDS; i++)
    if (strcmp(kCardIds[i], id) == 0) return i;
  return -1;
}
int monsoon_card_supported(int c) { return c >= 0 && c < NUM_CARDS && !card_unsupported(c); }

const char* monsoon_last_error(monsoon_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void monsoon_destroy(monsoon_t* h) {
  if (!h) return;
  hipSetDevice(h->device);
  void* ptrs[] = {h->b.state, h->b.rng_out, h->b.rng_mt, h->b.meta, h->b.weights, h->b.stats, h->b.scores, h->b.best, h->b.prof, h->b.pop, h->b.cfeat,
                  h->d_bytes, h->d_decks, h->d_factions, h->d_seeds, h->d_masks, h->d_i32, h->d_f64, h->d_p1, h->d_p2, h->d_int};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (auto& pr : h->pending) {
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int monsoon_create(const monsoon_config* cfg, monsoon_t** out) {
  if (!cfg || !out || cfg->max_games <= 0) {
    g_create_error = "monsoon_create: bad config";
    return MONSOON_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || cfg->device >= ndev) {
    g_create_error = "monsoon_create: no usable HIP device (this library has no CPU path)";
    return MONSOON_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) {
    g_create_error = "monsoon_create: hipGetDeviceProperties failed";
    return MONSOON_ERR_DEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("monsoon_create: built for gfx950, device is ") + prop.gcnArchName;
    return MONSOON_ERR_DEVICE;
  }
  monsoon* h = new monsoon();
  memset(&h->b, 0, sizeof(h->b));
  h->cfg = *cfg;
  if (h->cfg.lanes_per_game != 8 && h->cfg.lanes_per_game != 16 && h->cfg.lanes_per_game != 32 && h->cfg.lanes_per_game != 64)
    h->cfg.lanes_per_game = 8;
  h->wpe = h->cfg.lanes_per_game == 8 ? 4 : 2;
  if (const char* e = getenv("MONSOON_WPE")) h->wpe = atoi(e);   // tuning knob: min waves per SIMD the kernel is built for
  if (h->cfg.stack_bytes <= 0) h->cfg.stack_bytes = 16384;
  h->device = cfg->device;
  h->n = 0;
  h->n_individuals = 0;
  h->stream = nullptr;
  h->d_bytes = nullptr; h->d_decks = nullptr; h->d_factions = nullptr; h->d_seeds = nullptr; h->d_masks = nullptr;
  h->d_i32 = nullptr; h->d_f64 = nullptr; h->d_p1 = nullptr; h->d_p2 = nullptr; h->d_int = nullptr;
  h->kernel_ms = 0;
  h->kernel_launches = 0;
  h->parity = 0;
  h->grid_waves = 0;
  memset(h->st_acc, 0, sizeof(h->st_acc));
  memset(h->st_base, 0, sizeof(h->st_base));
  *out = h;
  size_t cap = (size_t)cfg->max_games;
  HIP_TRY(h, hipSetDevice(h->device));
  // the rules core recurses (move -> ability -> ...): give every lane a scratch stack
  HIP_TRY(h, hipDeviceSetLimit(hipLimitStackSize, (size_t)h->cfg.stack_bytes));
  HIP_TRY(h, hipStreamCreate(&h->stream));
  HIP_TRY(h, hipMalloc(&h->b.state, cap * SW * 4));
  HIP_TRY(h, hipMalloc(&h->b.rng_out, cap * RNG_WORDS * 4));
  HIP_TRY(h, hipMalloc(&h->b.rng_mt, cap * MT_N * 4));
  HIP_TRY(h, hipMalloc(&h->b.meta, cap * sizeof(GameMeta)));
  HIP_TRY(h, hipMemset(h->b.meta, 0, cap * sizeof(GameMeta)));
  HIP_TRY(h, hipMalloc(&h->b.stats, ST_WORDS * sizeof(unsigned long long)));
  HIP_TRY(h, hipMemset(h->b.stats, 0, ST_WORDS * sizeof(unsigned long long)));
  HIP_TRY(h, hipMalloc(&h->b.best, cap * sizeof(double)));
  HIP_TRY(h, hipMalloc(&h->b.cfeat, cap * MONSOON_NUM_ACTIONS * 10 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&h->b.pop, 2 * 8 * 32 * sizeof(int)));
  HIP_TRY(h, hipMemset(h->b.pop, 0, 2 * 8 * 32 * sizeof(int)));
#if defined(MSB_PROF) && MSB_PROF
  HIP_TRY(h, hipMalloc(&h->b.prof, cap * PROF_WORDS * sizeof(unsigned long long)));
  HIP_TRY(h, hipMemset(h->b.prof, 0, cap * PROF_WORDS * sizeof(unsigned long long)));
#endif
  HIP_TRY(h, hipMalloc(&h->d_decks, cap * 24));
  HIP_TRY(h, hipMalloc(&h->d_factions, cap * 2));
  HIP_TRY(h, hipMalloc(&h->d_seeds, cap * 4));
  HIP_TRY(h, hipMalloc(&h->d_masks, cap * 24));
  HIP_TRY(h, hipMalloc(&h->d_bytes, cap * 8 > 4096 ? cap * 8 : 4096));
  HIP_TRY(h, hipMalloc(&h->d_p1, cap * 4));
  HIP_TRY(h, hipMalloc(&h->d_p2, cap * 4));
  HIP_TRY(h, hipMalloc(&h->d_int, 64));
  return MONSOON_OK;
}

static int check_ready(monsoon_t* h) {
  if (!h) return MONSOON_ERR_ARG;
  if (h->n <= 0) {
    h->err = "no games loaded: call monsoon_reset first";
    return MONSOON_ERR_STATE;
  }
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) {
    h->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return MONSOON_ERR_DEVICE;
  }
  return MONSOON_OK;
}

static int launch_reset(monsoon_t* h, int n) {
  hipLaunchKernelGGL(k_seed, dim3(n), dim3(64), 0, h->stream, h->b, n, h->d_seeds);
  hipLaunchKernelGGL(k_init, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, h->d_decks, h->d_factions);
  HIP_TRY(h, hipGetLastError());
  return MONSOON_OK;
}

static int fold_stats(monsoon_t* h);

int monsoon_reset(monsoon_t* h, int32_t n, const uint32_t* seeds, const uint8_t* decks, const uint8_t* factions) {
  if (!h || !seeds || !decks || n <= 0 || n > h->cfg.max_games) {
    if (h) h->err = "monsoon_reset: bad argument";
    return MONSOON_ERR_ARG;
  }
  for (size_t i = 0; i < (size_t)n * 24; i++) {
    if (decks[i] >= NUM_CARDS || card_unsupported(decks[i])) {
      h->err = std::string("monsoon_reset: card not supported by this build: ") +
               (decks[i] < NUM_CARDS ? kCardIds[decks[i]] : "index out of range");
      return MONSOON_ERR_ARG;
    }
  }
  HIP_TRY(h, hipSetDevice(h->device));
  {
    int rc = fold_stats(h);   // statistics live in the per-game rows that are about to be cleared
    if (rc) return rc;
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_seeds, seeds, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_decks, decks, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
  if (factions)
    HIP_TRY(h, hipMemcpyAsync(h->d_factions, factions, (size_t)n * 2, hipMemcpyHostToDevice, h->stream));
  else
    HIP_TRY(h, hipMemsetAsync(h->d_factions, 0, (size_t)n * 2, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->b.meta, 0, (size_t)n * sizeof(GameMeta), h->stream));
  int rc = launch_reset(h, n);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n = n;
  return MONSOON_OK;
}

int monsoon_legal_mask(monsoon_t* h, uint64_t* out) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return MONSOON_ERR_ARG;
  int n = h->n;
  hipLaunchKernelGGL(k_legal, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, h->d_masks);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->d_masks, (size_t)n * 24, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_step(monsoon_t* h, const uint8_t* actions, int8_t* reward, uint8_t* done, uint8_t* fault) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!actions) return MONSOON_ERR_ARG;
  int n = h->n;
  uint8_t* d = h->d_bytes;   // [actions | reward | done | fault | illegal] x n
  HIP_TRY(h, hipMemcpyAsync(d, actions, n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_step, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, d, (int8_t*)(d + n), d + 2 * (size_t)n,
                     d + 3 * (size_t)n, d + 4 * (size_t)n);
  HIP_TRY(h, hipGetLastError());
  std::vector<uint8_t> host(4 * (size_t)n);
  HIP_TRY(h, hipMemcpyAsync(host.data(), d + n, 4 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; i++) {
    if (host[3 * (size_t)n + i]) {
      h->err = "monsoon_step: illegal action " + std::to_string(actions[i]) + " for game " + std::to_string(i);
      return MONSOON_ERR_ARG;
    }
  }
  if (reward) memcpy(reward, host.data(), n);
  if (done) memcpy(done, host.data() + n, n);
  if (fault) memcpy(fault, host.data() + 2 * (size_t)n, n);
  return MONSOON_OK;
}

int monsoon_expert_action(monsoon_t* h, uint8_t* out_action, uint8_t* fault) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out_action) return MONSOON_ERR_ARG;
  int n = h->n;
  uint8_t* d = h->d_bytes;
  hipLaunchKernelGGL(k_expert, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, d, d + n);
  HIP_TRY(h, hipGetLastError());
  std::vector<uint8_t> host(2 * (size_t)n);
  HIP_TRY(h, hipMemcpyAsync(host.data(), d, 2 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  memcpy(out_action, host.data(), n);
  if (fault) memcpy(fault, host.data() + n, n);
  return MONSOON_OK;
}

int monsoon_observe(monsoon_t* h, int32_t* out, uint8_t* raises) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return MONSOON_ERR_ARG;
  int n = h->n;
  if (!h->d_i32) HIP_TRY(h, hipMalloc(&h->d_i32, (size_t)h->cfg.max_games * MONSOON_OBS_INTS * 4));
  hipLaunchKernelGGL(k_observe, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, h->d_i32, h->d_bytes);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->d_i32, (size_t)n * MONSOON_OBS_INTS * 4, hipMemcpyDeviceToHost, h->stream));
  std::vector<uint8_t> r(n);
  HIP_TRY(h, hipMemcpyAsync(r.data(), h->d_bytes, n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (raises) memcpy(raises, r.data(), n);
  return MONSOON_OK;
}

// Device-pointer variant: writes (n,27,5,4) int32 straight into caller-owned DEVICE memory (e.g. a torch-ROCm
// tensor's data_ptr) -- no host round trip.  raises_dev (n bytes, device) may be NULL.
int monsoon_observe_dev(monsoon_t* h, void* out_dev, void* raises_dev) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out_dev) return MONSOON_ERR_ARG;
  int n = h->n;
  uint8_t* r = raises_dev ? (uint8_t*)raises_dev : h->d_bytes;
  hipLaunchKernelGGL(k_observe, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, (int32_t*)out_dev, r);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_features(monsoon_t* h, double* out) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return MONSOON_ERR_ARG;
  int n = h->n;
  if (!h->d_f64) HIP_TRY(h, hipMalloc(&h->d_f64, (size_t)h->cfg.max_games * 10 * 8));
  hipLaunchKernelGGL(k_features, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, h->d_f64);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->d_f64, (size_t)n * 80, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_status(monsoon_t* h, int32_t* out) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return MONSOON_ERR_ARG;
  int n = h->n;
  if (!h->d_i32) HIP_TRY(h, hipMalloc(&h->d_i32, (size_t)h->cfg.max_games * MONSOON_OBS_INTS * 4));
  hipLaunchKernelGGL(k_status, dim3((n + API_LANES - 1) / API_LANES), dim3(64), 0, h->stream, h->b, n, h->d_i32);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->d_i32, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_state_export(monsoon_t* h, int32_t idx, uint8_t* buf, int32_t* len) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!buf || !len || idx < 0 || idx >= h->n) return MONSOON_ERR_ARG;
  hipLaunchKernelGGL(k_export, dim3(1), dim3(64), API_LDS_BYTES, h->stream, h->b, idx, h->d_bytes, (int32_t*)(h->d_bytes + 2048));
  HIP_TRY(h, hipGetLastError());
  uint8_t host[2048 + 4];
  HIP_TRY(h, hipMemcpyAsync(host, h->d_bytes, sizeof(host), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  int32_t n;
  memcpy(&n, host + 2048, 4);
  memcpy(buf, host, n);
  *len = n;
  return MONSOON_OK;
}

// Raw record bytes of game idx as they sit in HBM (debugging aid; layout = state.h, not part of the parity surface).
int monsoon_debug_raw(monsoon_t* h, int32_t idx, uint8_t* buf, int32_t* len) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!buf || !len || idx < 0 || idx >= h->n) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(buf, h->b.state + (size_t)idx * SW, STATE_BYTES, hipMemcpyDeviceToHost));
  *len = STATE_BYTES;
  return MONSOON_OK;
}

int monsoon_game_faults(monsoon_t* h, uint8_t* out) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return MONSOON_ERR_ARG;
  int n = h->n;
  hipLaunchKernelGGL(k_faults, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, h->d_bytes);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->d_bytes, (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_state_hash(monsoon_t* h, uint64_t* out) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!out) return MONSOON_ERR_ARG;
  int n = h->n;
  hipLaunchKernelGGL(k_hash, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, h->d_masks);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(out, h->d_masks, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_upload_weights(monsoon_t* h, const double* weights, int32_t n_individuals) {
  if (!h || !weights || n_individuals <= 0) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  if (n_individuals > h->n_individuals) {
    if (h->b.weights) HIP_TRY(h, hipFree(h->b.weights));
    h->b.weights = nullptr;
    HIP_TRY(h, hipMalloc(&h->b.weights, (size_t)n_individuals * 80));
    h->n_individuals = n_individuals;
  }
  HIP_TRY(h, hipMemcpyAsync(h->b.weights, weights, (size_t)n_individuals * 80, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_assign_players(monsoon_t* h, const int32_t* p1, const int32_t* p2) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!p1 || !p2) return MONSOON_ERR_ARG;
  int n = h->n;
  for (int i = 0; i < n; i++)
    if (p1[i] < 0 || p2[i] < 0 || p1[i] >= h->n_individuals || p2[i] >= h->n_individuals) {
      h->err = "monsoon_assign_players: index outside the uploaded weight table";
      return MONSOON_ERR_ARG;
    }
  HIP_TRY(h, hipMemcpyAsync(h->d_p1, p1, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_p2, p2, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_assign, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, h->d_p1, h->d_p2);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

static int launch_decide(monsoon_t* h, int n, int max_turns, int write_scores, bool timed) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) {
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
  }
  static const int persistent = getenv("MONSOON_PERSIST") ? atoi(getenv("MONSOON_PERSIST")) : 1;
  static const int lds_pad = getenv("MONSOON_LDS_PAD") ? atoi(getenv("MONSOON_LDS_PAD")) : 0;   // occupancy experiments only
  if (timed) HIP_TRY(h, hipEventRecord(e0, h->stream));
#define MSB_LAUNCH(U, W)                                                                                                \
  do {                                                                                                                  \
    if (!h->grid_waves) {                                                                                               \
      int per_cu = 0;                                                                                                   \
      hipDeviceProp_t prop;                                                                                             \
      HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));                                                             \
      HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_decide<U, W>, 64, DecideLds<U>::TOTAL + lds_pad)); \
      h->grid_waves = 2 * (per_cu > 0 ? per_cu * prop.multiProcessorCount : 4096);   /* two waves per slot: measured best */                                            \
      if (const char* e = getenv("MONSOON_GRID")) h->grid_waves = atoi(e);                                              \
    }                                                                                                                   \
    int grid = (persistent && h->grid_waves < n) ? h->grid_waves : n;                                                   \
    hipLaunchKernelGGL((k_decide<U, W>), dim3(grid), dim3(64), DecideLds<U>::TOTAL + lds_pad, h->stream, h->b, n, max_turns, write_scores, persistent, h->parity); \
    h->parity ^= 1; \
  } while (0)
  int variant = h->cfg.lanes_per_game * 10 + h->wpe;
  switch (variant) {
    case 81: MSB_LAUNCH(8, 1); break;
    case 82: MSB_LAUNCH(8, 2); break;
    case 83: MSB_LAUNCH(8, 3); break;
    case 161: MSB_LAUNCH(16, 1); break;
    case 163: MSB_LAUNCH(16, 3); break;
    case 164: MSB_LAUNCH(16, 4); break;
#if !(defined(MSB_EXT) && MSB_EXT)
    case 321: MSB_LAUNCH(32, 1); break;
    case 322: MSB_LAUNCH(32, 2); break;
    case 641: MSB_LAUNCH(64, 1); break;
#endif
    case 162: MSB_LAUNCH(16, 2); break;
    default: MSB_LAUNCH(8, 4); break;
  }
#undef MSB_LAUNCH
  HIP_TRY(h, hipGetLastError());
  if (timed) {
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    h->pending.emplace_back(e0, e1);
  }
  return MONSOON_OK;
}

static int drain_timing(monsoon_t* h) {
  for (auto& pr : h->pending) {
    HIP_TRY(h, hipEventSynchronize(pr.second));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, pr.first, pr.second));
    h->kernel_ms += ms;
    h->kernel_launches++;
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  h->pending.clear();
  return MONSOON_OK;
}

int monsoon_decide_round_dev(monsoon_t* h) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!h->b.weights) {
    h->err = "monsoon_decide_round_dev: upload weights and assign players first";
    return MONSOON_ERR_STATE;
  }
  return launch_decide(h, h->n, 0x7fff, 0, true);
}

int monsoon_sync(monsoon_t* h) {
  if (!h) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return drain_timing(h);
}

int monsoon_decide(monsoon_t* h, const double* weights, uint8_t* out_action, double* out_score, double* out_scores) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!weights) return MONSOON_ERR_ARG;
  int n = h->n;
  // weights[n][2][10] -> table of 2n rows, game g plays rows 2g / 2g+1
  rc = monsoon_upload_weights(h, weights, 2 * n);
  if (rc) return rc;
  std::vector<int32_t> p1(n), p2(n);
  for (int i = 0; i < n; i++) {
    p1[i] = 2 * i;
    p2[i] = 2 * i + 1;
  }
  rc = monsoon_assign_players(h, p1.data(), p2.data());
  if (rc) return rc;
  if (out_scores) {
    if (!h->b.scores) HIP_TRY(h, hipMalloc(&h->b.scores, (size_t)h->cfg.max_games * MONSOON_NUM_ACTIONS * 8));
    size_t cnt = (size_t)n * MONSOON_NUM_ACTIONS;
    hipLaunchKernelGGL(k_clear_scores, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->b.scores, cnt);
  }
  rc = launch_decide(h, n, 0x7fff, out_scores ? 1 : 0, false);
  if (rc) return rc;
  std::vector<GameMeta> meta(n);
  HIP_TRY(h, hipMemcpyAsync(meta.data(), h->b.meta, (size_t)n * sizeof(GameMeta), hipMemcpyDeviceToHost, h->stream));
  if (out_score) HIP_TRY(h, hipMemcpyAsync(out_score, h->b.best, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  if (out_scores)
    HIP_TRY(h, hipMemcpyAsync(out_scores, h->b.scores, (size_t)n * MONSOON_NUM_ACTIONS * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (out_action)
    for (int i = 0; i < n; i++) out_action[i] = meta[i].last_action;
  return MONSOON_OK;
}

int monsoon_rollout(monsoon_t* h, const double* weights, int32_t n_individuals, const monsoon_match* matches,
                    int32_t n_matches, const uint8_t* deck_pairs, int32_t n_decks, int32_t max_turns,
                    int32_t* out_counts, int8_t* out_results, int32_t* out_steps) {
  if (!h || !weights || !matches || !deck_pairs || !out_counts || n_matches <= 0 || n_individuals <= 0 || max_turns <= 0 ||
      max_turns > 30000)
    return MONSOON_ERR_ARG;
  int rc = monsoon_upload_weights(h, weights, n_individuals);
  if (rc) return rc;
  int cap = h->cfg.max_games;
  int32_t* d_counts = nullptr;
  int8_t* d_results = nullptr;
  int32_t* d_steps = nullptr;
  HIP_TRY(h, hipMalloc(&d_counts, (size_t)n_individuals * 12));
  HIP_TRY(h, hipMemset(d_counts, 0, (size_t)n_individuals * 12));
  HIP_TRY(h, hipMalloc(&d_results, (size_t)n_matches));
  HIP_TRY(h, hipMalloc(&d_steps, (size_t)n_matches * 4));
  std::vector<uint32_t> seeds;
  std::vector<uint8_t> decks;
  std::vector<int32_t> p1, p2;
  for (int base = 0; base < n_matches; base += cap) {
    int n = n_matches - base < cap ? n_matches - base : cap;
    seeds.resize(n);
    decks.resize((size_t)n * 24);
    p1.resize(n);
    p2.resize(n);
    for (int i = 0; i < n; i++) {
      const monsoon_match& mm = matches[base + i];
      if (mm.p1 < 0 || mm.p1 >= n_individuals || mm.p2 < 0 || mm.p2 >= n_individuals || (int)mm.deck >= n_decks) {
        h->err = "monsoon_rollout: schedule entry out of range";
        hipFree(d_counts); hipFree(d_results); hipFree(d_steps);
        return MONSOON_ERR_ARG;
      }
      seeds[i] = mm.seed;
      memcpy(&decks[(size_t)i * 24], deck_pairs + (size_t)mm.deck * 24, 24);
      p1[i] = mm.p1;
      p2[i] = mm.p2;
    }
    rc = monsoon_reset(h, n, seeds.data(), decks.data(), nullptr);
    if (!rc) rc = monsoon_assign_players(h, p1.data(), p2.data());
    if (rc) {
      hipFree(d_counts); hipFree(d_results); hipFree(d_steps);
      return rc;
    }
    hipLaunchKernelGGL(k_set_match, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, base);
    // max_turns decision rounds + one closing round that turns "still running" into a result.
    // Every 16 rounds the live count is read back so that finished batches stop early.
    for (int round = 0; round <= max_turns; round++) {
      rc = launch_decide(h, n, max_turns, 0, true);
      if (rc) break;
      if ((round & 15) == 15) {
        int live = 0;
        HIP_TRY(h, hipMemsetAsync(h->d_int, 0, 4, h->stream));
        hipLaunchKernelGGL(k_count_live, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, h->d_int);
        HIP_TRY(h, hipMemcpyAsync(&live, h->d_int, 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        drain_timing(h);
        if (live == 0) break;
      }
    }
    if (rc) {
      hipFree(d_counts); hipFree(d_results); hipFree(d_steps);
      return rc;
    }
    hipLaunchKernelGGL(k_collect, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, d_counts, d_results, d_steps);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_timing(h);
  }
  std::vector<int32_t> counts((size_t)n_individuals * 3);
  HIP_TRY(h, hipMemcpy(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < counts.size(); i++) out_counts[i] += counts[i];
  if (out_results) HIP_TRY(h, hipMemcpy(out_results, d_results, (size_t)n_matches, hipMemcpyDeviceToHost));
  if (out_steps) HIP_TRY(h, hipMemcpy(out_steps, d_steps, (size_t)n_matches * 4, hipMemcpyDeviceToHost));
  hipFree(d_counts);
  hipFree(d_results);
  hipFree(d_steps);
  return MONSOON_OK;
}

// current statistics of the loaded games (see k_stats)
static int reduce_stats(monsoon_t* h, unsigned long long cur[5]) {
  for (int i = 0; i < 5; i++) cur[i] = 0;
  if (h->n <= 0) return MONSOON_OK;
  HIP_TRY(h, hipMemsetAsync(h->b.stats, 0, 5 * sizeof(unsigned long long), h->stream));
  int blocks = (h->n + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(k_stats, dim3(blocks), dim3(256), 0, h->stream, h->b, h->n, h->b.stats);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(cur, h->b.stats, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}
static int total_stats(monsoon_t* h, unsigned long long tot[5]) {
  unsigned long long cur[5];
  int rc = reduce_stats(h, cur);
  if (rc) return rc;
  for (int i = 0; i < 5; i++) tot[i] = h->st_acc[i] + cur[i] - h->st_base[i];
  return MONSOON_OK;
}
// the loaded games are about to be replaced: keep what they contributed
static int fold_stats(monsoon_t* h) {
  unsigned long long tot[5];
  int rc = total_stats(h, tot);
  if (rc) return rc;
  for (int i = 0; i < 5; i++) {
    h->st_acc[i] = tot[i];
    h->st_base[i] = 0;
  }
  return MONSOON_OK;
}

int monsoon_get_stats(monsoon_t* h, monsoon_stats* out) {
  if (!h || !out) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  unsigned long long s[5];
  int rc = total_stats(h, s);
  if (rc) return rc;
  out->lookahead_steps = s[ST_LOOKAHEAD];
  out->decisions = s[ST_DECISIONS];
  out->games_finished = s[ST_FINISHED];
  out->faults = s[ST_FAULTS];
  out->capacity_faults = s[ST_CAPFAULTS];
  return MONSOON_OK;
}

// Raw counter words; out = 192 u64 (profiling builds: k_decide phase cycles at 8.., function scopes at 32.. / 64..).
int monsoon_debug_counters(monsoon_t* h, unsigned long long* out) {
  if (!h || !out) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  memset(out, 0, 192 * sizeof(unsigned long long));
  {
    int rc = total_stats(h, out);
    if (rc) return rc;
  }
#if defined(MSB_PROF) && MSB_PROF
  {
    std::vector<unsigned long long> v((size_t)h->cfg.max_games * PROF_WORDS);
    HIP_TRY(h, hipMemcpy(v.data(), h->b.prof, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < v.size(); i++) {
      size_t k = i % PROF_WORDS;
      if (k < 72) out[k < 8 ? ST_PROF + k : 32 + (k - 8)] += v[i];   // 32..63 scope cycles, 64..95 scope calls
      else if (k < 136) out[128 + (k - 72)] += v[i];                 // 128..159 call-entry cycles, 160..191 call-exit cycles
    }
    // occupancy of the LAST launch from the per-wave wall-clock stamps (100 MHz): span, sum of wave times,
    // time at which the 4096th-from-last wave ended (start of the tail)
    {
      std::vector<unsigned long long> st, en;
      unsigned long long newest = 0;
      for (int g = 0; g < h->n; g++) newest = std::max(newest, v[(size_t)g * PROF_WORDS + 137]);
      for (int g = 0; g < h->n; g++) {
        unsigned long long a = v[(size_t)g * PROF_WORDS + 136], e = v[(size_t)g * PROF_WORDS + 137];
        if (e > a && e + 1000000ull > newest) {   // stamps of the last launch only (within 10 ms of the newest)
          st.push_back(a);
          en.push_back(e);
        }
      }
      if (!st.empty()) {
        unsigned long long t0 = *std::min_element(st.begin(), st.end()), t1 = *std::max_element(en.begin(), en.end());
        unsigned long long sum = 0, longest = 0;
        for (size_t i = 0; i < st.size(); i++) {
          sum += en[i] - st[i];
          longest = std::max(longest, en[i] - st[i]);
        }
        std::sort(en.begin(), en.end());
        std::sort(st.begin(), st.end());
        out[96] = t1 - t0;
        out[97] = sum;
        out[98] = en.size() > 4096 ? en[en.size() - 4096] - t0 : 0;
        out[99] = longest;
        out[100] = st.size();
        out[101] = st.back() - t0;   // when the last wave started
      }
    }
  }
#endif
  return MONSOON_OK;
}

int monsoon_reset_stats(monsoon_t* h) {
  if (!h) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  drain_timing(h);
  {
    unsigned long long cur[5];
    int rc = reduce_stats(h, cur);
    if (rc) return rc;
    for (int i = 0; i < 5; i++) {
      h->st_acc[i] = 0;
      h->st_base[i] = cur[i];
    }
  }
  if (h->b.prof) HIP_TRY(h, hipMemset(h->b.prof, 0, (size_t)h->cfg.max_games * PROF_WORDS * sizeof(unsigned long long)));
  h->kernel_ms = 0;
  h->kernel_launches = 0;
  return MONSOON_OK;
}

int monsoon_kernel_time(monsoon_t* h, double* total_ms, int64_t* launches) {
  if (!h) return MONSOON_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  int rc = drain_timing(h);
  if (rc) return rc;
  if (total_ms) *total_ms = h->kernel_ms;
  if (launches) *launches = h->kernel_launches;
  return MONSOON_OK;
}

void* monsoon_stream(monsoon_t* h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"
