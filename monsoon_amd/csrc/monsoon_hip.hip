// libmonsoon_hip.so -- MI355X (gfx950) batched Stormbound engine: API kernels + C ABI (include/monsoon.h).
//
// The hot kernel (k_play, kernels.h) is compiled one variant per translation unit (variant.hip); this
// file holds the lane-per-game API kernels (reset/step/legal/observe/features/status/export: one LANE per game with
// the same lane-interleaved LDS layout; they back the batch=1 Game view and the parity tests) and the host side.
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

#include "kernels.h"

using namespace msb;
using namespace msbk;

namespace {

// ------------------------------------------------------------------------------------------------
// Lane-per-game API kernels.  Block = 64 threads; the records are staged in DYNAMIC LDS starting at LDS
// address 0 (these kernels declare no static __shared__), interleaved across lanes in 16-byte granules.
// ------------------------------------------------------------------------------------------------
// games per 64-thread block in the API kernels: the extended record is too large for 64 LDS columns
#if defined(MSB_API_LANES)
constexpr int API_LANES = MSB_API_LANES;   // probes only (scripts/probe/divergence2.sh)
#elif defined(MSB_EXT) && MSB_EXT == 2
constexpr int API_LANES = 8;
#elif defined(MSB_EXT) && MSB_EXT
constexpr int API_LANES = 16;
#else
constexpr int API_LANES = 64;
#endif
// LDS address 0 is avoided on purpose: an integer constant 0 cast to an LDS pointer is the null pointer,
// which is not address 0 on this target; every region starts at LDS_ORIGIN.
constexpr int API_SKB = LDS_ORIGIN + SG * API_LANES * 16;   // the lanes' work stacks behind their records
constexpr int API_LDS_BYTES = API_SKB + API_LANES * SKW * 4;
typedef LaneMem<API_LANES, LDS_ORIGIN, API_SKB, SKW> ApiMem;
typedef Engine<ApiMem> ApiEngine;
#define API_GAME_INDEX()                                  \
  lds_init_wtab(b.wk_ovf + (size_t)blockIdx.x * (API_LANES * OVF_WORDS)); \
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

__global__ void __launch_bounds__(64) k_seed(DevBuffers b, int g0, int n, const uint32_t* seeds) {
  // one wavefront per game (games g0 .. g0+n-1, seeds[0..n-1]): init_genrand is a serial recurrence (lane 0), the two
  // twists are wave-cooperative
  __shared__ uint32_t tmp[MT_N];
  int lane = threadIdx.x;
  if ((int)blockIdx.x >= n) return;
  int g = g0 + blockIdx.x;
  seeds -= g0;
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

__global__ void __launch_bounds__(64) k_init(DevBuffers b, int n, const uint8_t* decks, const uint8_t* factions, const uint32_t* seeds) {
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
  e.init_game(d0, d1, factions[2 * g], factions[2 * g + 1], seeds[g]);
  if (e.rng_pos() >= (uint32_t)MT_N) e.rng_block_advance();
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
  if (e.rng_pos() >= (uint32_t)MT_N) e.rng_block_advance();
  lane_commit_rng(b, g, m, e.rng_pos());
  m.steps++;
  m.last_action = (uint8_t)a;
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
  if (e.rng_pos() >= (uint32_t)MT_N) {   // the record counts its stream blocks (extended build: state.h X_RNGBLK)
    e.rng_block_advance();
    api_store(b.state + (size_t)g * SW);
  }
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
  lds_init_wtab();
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

// Statistics of the loaded games, reduced from the per-game fields: {look-ahead steps, decisions, games ended by a
// winner, games stopped by a fault, of those build-limit faults, games with a build-limit fault inside a look-ahead}.
// One atomic per wavefront into a zeroed buffer.
__global__ void __launch_bounds__(256) k_stats(DevBuffers b, int n, unsigned long long* out) {
  unsigned long long v[ST_N] = {0, 0, 0, 0, 0, 0};
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
    GameMeta m = b.meta[g];
    v[0] += m.lookahead;
    v[1] += m.decided;
    v[2] += (m.flags & 1) ? 1 : 0;
    v[3] += (m.result == -1 && m.fault) ? 1 : 0;
    v[4] += (m.result == -1 && m.fault >= FAULT_CAPACITY) ? 1 : 0;
    v[5] += m.la_fault ? 1 : 0;
  }
  for (int i = 0; i < ST_N; i++) {
    unsigned long long x = v[i];
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out[i], x);
  }
}

// per game: a build-limit code (>= FAULT_CAPACITY) whenever the game met one -- as the fault that stopped it or inside a
// look-ahead (that action scored 0.0 where the reference computes a score, so the game may have left the reference's
// line even if it later ended with one of the reference's own exceptions) -- else the fault that stopped it (0 = none)
__device__ inline int reported_fault(const GameMeta& m) {
  return m.fault >= FAULT_CAPACITY ? m.fault : (m.la_fault ? m.la_fault : m.fault);
}
__global__ void k_faults(DevBuffers b, int n, uint8_t* out) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = (uint8_t)reported_fault(b.meta[g]);
}

__global__ void k_clear_scores(double* scores, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scores[i] = NAN;
}

__global__ void k_assign(DevBuffers b, int n, const int32_t* p1, const int32_t* p2, int match_base) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  b.meta[g].p1 = p1[g];
  b.meta[g].p2 = p2[g];
  b.meta[g].match = (uint32_t)(match_base + g);
}

__global__ void k_collect(DevBuffers b, int n, int32_t* counts, int8_t* results, int32_t* steps, uint8_t* faults) {
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
  if (faults) faults[m.match] = (uint8_t)reported_fault(m);
}

// monsoon_debug_build / monsoon_debug_op (diagnostics; scenario tests): ONE lane runs scenario.inc on game g.  The
// engine of this kernel logs the order in which abilities run (TraceLaneMem) -- the product kernels' engines do not.
constexpr int DBG_TRACE_CAP = 256;
constexpr int DBG_TRACE = LDS_ORIGIN + SG * 16;                      // u32 count, then {card, position} pairs
constexpr int DBG_SKB = DBG_TRACE + 4 + 8 * DBG_TRACE_CAP;          // the one lane's work stack
constexpr int DBG_LDS_BYTES = DBG_SKB + SKW * 4;
typedef Engine<TraceLaneMem<1, LDS_ORIGIN, DBG_SKB, SKW, DBG_TRACE, DBG_TRACE_CAP>> DbgEngine;
__global__ void __launch_bounds__(64) k_debug(DevBuffers b, int g, int build, uint32_t seed, uint32_t stream_pos, const int32_t* stream,
                                               int32_t* out /* fault, n_log, pairs... */) {
  lds_init_wtab(b.wk_ovf);   // one lane, one block: the first words of the overflow buffer
  if (threadIdx.x != 0) return;
  DbgEngine e;
  MSB_AS_LDS int32_t* tr = (MSB_AS_LDS int32_t*)(uintptr_t)DBG_TRACE;
  tr[0] = 0;
  MSB_AS_LDS u32x4* rec = (MSB_AS_LDS u32x4*)(uintptr_t)LDS_ORIGIN;
  u32x4* grec = (u32x4*)(b.state + (size_t)g * SW);
  GameMeta m = b.meta[g];
  int fault = 0;
  if (build) {
    // the stream was seeded by k_seed (cursor 0 of block 0): move it to stream_pos
    m = GameMeta{};
    m.rng = 0;
    uint32_t blocks = stream_pos / (uint32_t)MT_N;
    for (uint32_t k = 0; k < blocks; k++) lane_commit_rng(b, g, m, (uint32_t)MT_N);
    m.rng = (m.rng & 0x10000u) | (stream_pos % (uint32_t)MT_N);
    for (int c = 0; c < SG; c++) rec[c] = u32x4{0u, 0u, 0u, 0u};
    attach_rng(e, b, g, m.rng);
    if (REM_LISTS) e.m.st16(X_RNGBLK, (int)blocks);
    e.scn_build(stream, seed);
    fault = e.fault();
    m.result = -2;
    m.last_action = 255;
  } else {
    for (int c = 0; c < SG; c++) rec[c] = grec[c];
    attach_rng(e, b, g, m.rng);
    fault = e.scn_op(stream);
    if (e.rng_pos() >= (uint32_t)MT_N) e.rng_block_advance();
    lane_commit_rng(b, g, m, e.rng_pos());
  }
  for (int c = 0; c < SG; c++) grec[c] = rec[c];
  b.meta[g] = m;
  out[0] = fault;
  int n = tr[0];
  out[1] = n;
  for (int i = 0; i < 2 * n; i++) out[2 + i] = tr[1 + i];
}

// monsoon_debug_kat (diagnostics): the device's numpy-stream draws and its score arithmetic on caller-given inputs, so
// that the known-answer vectors generated from numpy itself (tests/golden/rng_kat.npz, score_kat.npz) can be put to the
// HIP code directly.  One lane; the draws come from game 0's stream buffers, freshly seeded by the host side.
__global__ void __launch_bounds__(64) k_kat(DevBuffers b, int kind, int n, const void* in, void* out) {
  lds_init_wtab();
  if (threadIdx.x != 0) return;
  DbgEngine e;
  MSB_AS_LDS u32x4* rec = (MSB_AS_LDS u32x4*)(uintptr_t)LDS_ORIGIN;
  for (int c = 0; c < SG; c++) rec[c] = u32x4{0u, 0u, 0u, 0u};
  GameMeta m = GameMeta{};
  attach_rng(e, b, 0, m.rng);
  auto commit = [&]() {   // keep the window ahead of the cursor, as every committed step does
    if (e.rng_pos() >= (uint32_t)MT_N) {
      lane_commit_rng(b, 0, m, e.rng_pos());
      attach_rng(e, b, 0, m.rng);
    }
  };
  for (int i = 0; i < n; i++) {
    if (kind == 0) ((uint32_t*)out)[i] = e.rng_next_u32();
    else if (kind == 1) ((double*)out)[i] = e.rng_random_sample();
    else if (kind == 2) ((int32_t*)out)[i] = e.rng_randint(0, ((const int32_t*)in)[i]);
    else if (kind == 3) {   // shuffle(list(range(12)))
      int32_t* a = (int32_t*)out + 12 * i;
      for (int k = 0; k < 12; k++) a[k] = k;
      for (int k = 11; k >= 1; k--) {
        int j = (int)e.rng_interval((uint32_t)k);
        int t = a[k];
        a[k] = a[j];
        a[j] = t;
      }
    } else {                // score of (w[10], before[10], after[10])
      const double* r = (const double*)in + 30 * i;
      double w[10], fb[10], fa[10];
      for (int k = 0; k < 10; k++) {
        w[k] = r[k];
        fb[k] = r[10 + k];
        fa[k] = r[20 + k];
      }
      ((double*)out)[i] = DbgEngine::action_score(w, fb, fa);
    }
    commit();
  }
}

// monsoon_state_save / monsoon_state_load: one game's complete device state as a flat blob
// {u32 magic, u32 record bytes, GameMeta, record, raw MT state, two tempered blocks}
constexpr uint32_t BLOB_MAGIC = 0x4d53424cu ^ (uint32_t)STATE_BYTES;
constexpr int BLOB_BYTES = 8 + (int)sizeof(GameMeta) + STATE_BYTES + MT_N * 4 + RNG_WORDS * 4;
__global__ void k_blob(DevBuffers b, int g, uint32_t* blob, int load) {
  const int t = threadIdx.x;
  uint32_t* meta = (uint32_t*)(b.meta + g);
  uint32_t* parts[4] = {meta, b.state + (size_t)g * SW, b.rng_mt + (size_t)g * MT_N, b.rng_out + (size_t)g * RNG_WORDS};
  const int words[4] = {(int)sizeof(GameMeta) / 4, SW, MT_N, RNG_WORDS};
  int off = 2;
  if (!load && t == 0) {
    blob[0] = BLOB_MAGIC;
    blob[1] = (uint32_t)STATE_BYTES;
  }
  for (int p = 0; p < 4; p++) {
    for (int i = t; i < words[p]; i += blockDim.x) {
      if (load) parts[p][i] = blob[off + i];
      else blob[off + i] = parts[p][i];
    }
    off += words[p];
  }
}

// monsoon_draw_decks: numpy.random.RandomState(seed).choice(pool, 12, replace=False), twice, for every seed -- the
// per-game decks of configuration C5 (two draws from a pre-stream of the game's own, seeded with
// seed ^ 0x9E3779B9 by the caller).  Legacy choice without replacement is permutation(len(pool))[:12] (numpy
// mtrand.pyx: choice -> permutation -> shuffle -> _shuffle_raw): a Fisher-Yates shuffle of arange(len(pool)) from the top
// down with j = random_interval(i) (masked rejection, mt19937.h / Appendix C), the same shuffle Player.__init__ applies
// to a deck (player.py:28).  One wavefront per seed: init_genrand is a serial recurrence (lane 0), the two twists that
// yield the first 1 248 outputs are wave-cooperative, the 2 x (len(pool) - 1) draws serial again.  A seed whose draws
// need more than 1 248 outputs (expected: 270) is reported, never truncated.
__global__ void __launch_bounds__(64) k_draw_decks(int n, const uint32_t* seeds, const uint8_t* pool, int pool_n, uint8_t* out, int* overrun) {
  __shared__ uint32_t mt[MT_N];
  __shared__ uint32_t words[2 * MT_N];
  __shared__ uint8_t perm[128];
  const int lane = threadIdx.x, g = blockIdx.x;
  if (g >= n) return;
  MSB_AS_LDS uint32_t* t = (MSB_AS_LDS uint32_t*)mt;
  if (lane == 0) {
    uint32_t x = seeds[g];
    t[0] = x;
    for (int i = 1; i < MT_N; i++) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      t[i] = x;
    }
  }
  __syncthreads();
  for (int blk = 0; blk < 2; blk++) {
    wave_twist_lds(t, lane);
    for (int k = lane; k < MT_N; k += 64) words[blk * MT_N + k] = mt_temper(t[k]);
    __syncthreads();
  }
  if (lane != 0) return;
  int pos = 0;
  bool over = false;
  for (int side = 0; side < 2; side++) {
    for (int i = 0; i < pool_n; i++) perm[i] = (uint8_t)i;
    for (int i = pool_n - 1; i >= 1; i--) {
      uint32_t mask = (uint32_t)i;
      mask |= mask >> 1;
      mask |= mask >> 2;
      mask |= mask >> 4;
      uint32_t v = 0;
      do {
        if (pos >= 2 * MT_N) {
          over = true;
          v = 0;
          break;
        }
        v = words[pos++] & mask;
      } while (v > (uint32_t)i);
      uint8_t tmp = perm[i];
      perm[i] = perm[v];
      perm[v] = tmp;
    }
    for (int k = 0; k < 12; k++) out[(size_t)g * 24 + side * 12 + k] = pool[perm[k]];
  }
  if (over) atomicAdd(overrun, 1);
}

// ------------------------------------------------------------------------------------------------
// GA operators.  The reference's (mu + lambda) step draws everything from ONE global numpy
// stream (evo/population.py:75-89, evo/weights.py:12-40): per offspring randint(0, mu) picks the parent,
// WeightVector.copy() constructs a fresh vector first -- `dim` uniforms that are thrown away --, then mutate() draws one
// global normal, `dim` per-gene normals and `dim` steps.  The stream position after one offspring depends on its
// rejections, so the operator is serial by nature: ONE lane walks the stream here (the other 63 idle; at lambda = 4 096
// that is ~250 000 outputs).  What is bit-identical to numpy: every draw and every accept / reject decision, i.e. the
// parents chosen and the state handed back (key, position, the cached second normal up to the last bits of log / sqrt).
// What is not: exp / log / sqrt are the device library's (within 1 ulp of glibc's / numpy's SIMD exp): offspring agree
// with the host's to a few ulp.  The host GA stays the default and the bit-exact path.
// ------------------------------------------------------------------------------------------------
struct NpStream {   // numpy's legacy RandomState over an mt19937 state held in LDS
  MSB_AS_LDS uint32_t* key;
  int pos;
  int has_gauss;
  double gauss;
  long long tries;   // polar-method rounds so far (diagnostics: the accept / reject pattern)
  __device__ uint32_t u32() {   // mt19937_next: regenerate at position 624, temper on the way out
    if (pos == MT_N) {
      int k;
      for (k = 0; k < MT_N - MT_M; k++) key[k] = key[k + MT_M] ^ mt_mix(key[k], key[k + 1]);
      for (; k < MT_N - 1; k++) key[k] = key[k + (MT_M - MT_N)] ^ mt_mix(key[k], key[k + 1]);
      key[MT_N - 1] = key[MT_M - 1] ^ mt_mix(key[MT_N - 1], key[0]);
      pos = 0;
    }
    return mt_temper(key[pos++]);
  }
  __device__ double next_double() {   // legacy_double
    uint32_t a = u32() >> 5, b = u32() >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
  }
  __device__ uint32_t interval(uint32_t max) {   // random_interval / buffered_bounded_masked_uint32
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
      v = u32() & mask;
    } while (v > max);
    return v;
  }
  __device__ double gauss_next() {   // legacy_gauss (numpy/random/src/legacy/legacy-distributions.c): polar Box-Muller, one value cached
    if (has_gauss) {
      const double t = gauss;
      has_gauss = 0;
      gauss = 0.0;
      return t;
    }
    double f, x1, x2, r2;
    do {
      x1 = 2.0 * next_double() - 1.0;
      x2 = 2.0 * next_double() - 1.0;
      r2 = x1 * x1 + x2 * x2;
      tries++;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
};
__global__ void __launch_bounds__(64) k_ga_offspring(uint32_t* key_io, int* pos_gauss_io /* pos, has_gauss */, double* gauss_io, const double* pw,
                                                     const double* ps, int mu, int dim, int lambda, double tau, double tau_prime, double min_sigma,
                                                     double* out_w, double* out_s, int* out_parent, long long* out_tries) {
  __shared__ uint32_t key[MT_N];
  for (int k = threadIdx.x; k < MT_N; k += 64) key[k] = key_io[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    NpStream rs{(MSB_AS_LDS uint32_t*)key, pos_gauss_io[0], pos_gauss_io[1], gauss_io[0], 0};
    for (int c = 0; c < lambda; c++) {
      const int p = (int)rs.interval((uint32_t)(mu - 1));         // np.random.randint(0, mu)
      for (int i = 0; i < dim; i++) (void)rs.next_double();        // parent.copy(): WeightVector(size) draws uniform(0, 1, size)
      const double g = rs.gauss_next();                            // np.random.normal(0, 1)
      double sig[16];
      for (int i = 0; i < dim; i++) {                              // np.random.normal(0, 1, n); sigma' = max(sigma * exp(tau' g + tau z), eps)
        const double z = 0.0 + 1.0 * rs.gauss_next();
        const double a = tau_prime * g, b = tau * z;
        double sv = ps[(size_t)p * dim + i] * exp(a + b);
        sig[i] = sv > min_sigma ? sv : min_sigma;                  // np.maximum
        if (sv != sv) sig[i] = sv;
      }
      for (int i = 0; i < dim; i++) {                              // w' = clip(w + normal(0, sigma'), 0, 1)
        const double stepv = 0.0 + sig[i] * rs.gauss_next();
        double w = pw[(size_t)p * dim + i] + stepv;
        w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
        out_w[(size_t)c * dim + i] = w;
        out_s[(size_t)c * dim + i] = sig[i];
      }
      out_parent[c] = p;
      out_tries[c] = rs.tries;
    }
    pos_gauss_io[0] = rs.pos;
    pos_gauss_io[1] = rs.has_gauss;
    gauss_io[0] = rs.gauss;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < MT_N; k += 64) key_io[k] = key[k];
}
// (mu + lambda) selection, evo/population.py:91-103: sorted(zip(fitness, individuals), key=fitness, reverse=True)[:mu] -- CPython's
// sort is stable and reverse keeps equal keys in their original order, so the rank of i is the number of j with a larger
// fitness, or an equal one and a smaller index.  One thread per individual, O(n^2) compares (n <= a few thousand).
__global__ void k_ga_select(const double* fitness, int n, int* out_order) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double f = fitness[i];
  int rank = 0;
  for (int j = 0; j < n; j++) {
    const double g = fitness[j];
    rank += (g > f || (g == f && j < i)) ? 1 : 0;
  }
  out_order[rank] = i;
}

}  // namespace

// ================================================================================================
// Host side
// ================================================================================================
struct monsoon {
  monsoon_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuffers b;
  const VariantOps* var = nullptr;   // hot-kernel variant: candidate lanes per game, waves per SIMD
  int parity = 0;     // which of b.pop the next k_decide launch uses
  size_t ovf_lanes = 0;   // stepping lanes b.wk_ovf has room for (OVF_WORDS words each)
  unsigned long long st_acc[ST_N], st_base[ST_N];   // statistics: totals of earlier batches, baseline of the loaded one
  int grid_waves = 0; // persistent grid size of k_decide (resident wavefronts), 0 = not yet queried
  int n = 0;          // games loaded by the last reset
  int n_individuals = 0;   // rows of the uploaded weight table
  int weights_cap = 0;     // rows allocated
  std::string err;
  // scratch device buffers for API calls
  uint8_t* d_bytes = nullptr;   // max(cap * 8, 16 KiB)
  uint8_t* d_decks = nullptr;   // [cap][24]
  uint8_t* d_factions = nullptr;
  uint32_t* d_seeds = nullptr;
  uint64_t* d_masks = nullptr;
  int32_t* d_i32 = nullptr;     // cap * 540
  double* d_f64 = nullptr;      // cap * 10
  int32_t* d_p1 = nullptr;
  int32_t* d_p2 = nullptr;
  int* d_int = nullptr;
  // rollout result buffers, grown on demand and kept for the life of the handle
  int32_t* d_counts = nullptr;
  size_t counts_cap = 0;
  int8_t* d_results = nullptr;    // [matches_cap] results, then [matches_cap] fault codes (monsoon_rollout_faults)
  size_t rollout_matches = 0;     // matches of the last monsoon_rollout
  bool own_stream = false;
  int32_t* d_steps = nullptr;
  size_t matches_cap = 0;
  // kernel timing: event pairs are created once and reused
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double kernel_ms = 0;
  long long kernel_launches = 0;
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

// k_play instantiations of this build (variants.def): each one is a full compilation of the rules core
// in a translation unit of its own; anything else is refused by monsoon_create.
#include "variants.def"
#define X(U, W, G) const VariantOps* monsoon_variant_##U##_##W##_##G();
MSB_VARIANTS(X)
#undef X
static const VariantOps* find_variant(int u, int w, int g) {   // g: the kind of kernel (variants.def); w = 0: the one with the most waves per SIMD
  const VariantOps* best = nullptr;
#define X(U, W, G) if (u == U && g == G && (w == W || (w == 0 && (!best || W > best->wpe)))) best = monsoon_variant_##U##_##W##_##G();
  MSB_VARIANTS(X)
#undef X
  return best;
}
static const VariantOps* default_variant() {
  const VariantOps* first = nullptr;
#define X(U, W, G) if (!first) first = monsoon_variant_##U##_##W##_##G();
  MSB_VARIANTS(X)
#undef X
  return first;
}

extern "C" {

int monsoon_version(void) {
#if defined(MSB_EXT) && MSB_EXT == 2
  return 0x30002;   // bits 16 + 17: the large extended record (254 entity slots)
#elif defined(MSB_EXT) && MSB_EXT
  return 0x10002;   // bit 16: extended record
#else
  return 2;
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
  if (h->stream) hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->b.state, h->b.rng_out, h->b.rng_mt, h->b.meta, h->b.weights, h->b.stats, h->b.scores, h->b.best, h->b.prof, h->b.pop, h->b.wk_ovf,
                  h->d_bytes, h->d_decks, h->d_factions, h->d_seeds, h->d_masks, h->d_i32, h->d_f64, h->d_p1, h->d_p2, h->d_int,
                  h->d_counts, h->d_results, h->d_steps};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (auto& pr : h->ev_pool) {
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  if (h->stream && h->own_stream) hipStreamDestroy(h->stream);
  delete h;
}

static hipError_t bind_device(monsoon_t* h);
static int create_impl(monsoon* h) {
  const monsoon_config& cfg = h->cfg;
  size_t cap = (size_t)cfg.max_games;
  HIP_TRY(h, bind_device(h));
  // One stream per handle: independent handles do not serialise against each other or against the other blocking
  // streams of the process (a torch consumer of the observation tensor, say).  Round 2 had to put every handle on the
  // device's default stream: its kernels kept a 16-32 KiB per-lane stack in scratch memory and two hardware queues
  // holding GBs of scratch made the runtime hand it back and forth (200-500 ms per launch).  The rules core has no
  // stack in scratch any more (< 1 KiB of spill slots per lane): three builds alive on three streams launch in
  // 0.15-0.4 ms (scripts/probe/own_stream.py, profiles/r03_own_stream.txt).  MONSOON_OWN_STREAM=0 = the default stream.
  if (!getenv("MONSOON_OWN_STREAM") || atoi(getenv("MONSOON_OWN_STREAM")) != 0) {
    HIP_TRY(h, hipStreamCreate(&h->stream));
    h->own_stream = true;
  } else {
    h->stream = nullptr;
    h->own_stream = false;
  }
  HIP_TRY(h, hipMalloc(&h->b.state, cap * SW * 4));
  HIP_TRY(h, hipMalloc(&h->b.rng_out, cap * RNG_WORDS * 4));
  HIP_TRY(h, hipMalloc(&h->b.rng_mt, cap * MT_N * 4));
  HIP_TRY(h, hipMalloc(&h->b.meta, cap * sizeof(GameMeta)));
  HIP_TRY(h, hipMemset(h->b.meta, 0, cap * sizeof(GameMeta)));
  HIP_TRY(h, hipMalloc(&h->b.stats, ST_WORDS * sizeof(unsigned long long)));
  HIP_TRY(h, hipMemset(h->b.stats, 0, ST_WORDS * sizeof(unsigned long long)));
  HIP_TRY(h, hipMalloc(&h->b.best, cap * sizeof(double)));
  // overflow blocks of the work stack (kernels.h): the lane-per-game API kernels step up to `cap` games at once (rounded
  // up to whole workgroups); launch_play grows the buffer once it knows its grid
  h->ovf_lanes = ((cap + API_LANES - 1) / API_LANES) * (size_t)API_LANES;
  HIP_TRY(h, hipMalloc(&h->b.wk_ovf, h->ovf_lanes * OVF_WORDS * 4));
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
  HIP_TRY(h, hipMalloc(&h->d_bytes, cap * 8 > 16384 ? cap * 8 : 16384));
  HIP_TRY(h, hipMalloc(&h->d_p1, cap * 4));
  HIP_TRY(h, hipMalloc(&h->d_p2, cap * 4));
  HIP_TRY(h, hipMalloc(&h->d_int, 64));
  return MONSOON_OK;
}

int monsoon_create(const monsoon_config* cfg, monsoon_t** out) {
  if (out) *out = nullptr;
  if (!cfg || !out || cfg->max_games <= 0) {
    g_create_error = "monsoon_create: bad config";
    return MONSOON_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || cfg->device < 0 || cfg->device >= ndev) {
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
  // kernel variant: 0 = this build's default; an explicit value without an instantiation is an error, never a
  // silent substitute (tuning knobs for experiments: MONSOON_LANES / MONSOON_WPE)
  int u = cfg->lanes_per_game, w = 0, gpw = 0;
  if (const char* e = getenv("MONSOON_LANES")) u = atoi(e);
  if (const char* e = getenv("MONSOON_WPE")) w = atoi(e);
  if (const char* e = getenv("MONSOON_GAMES_PER_WAVE")) gpw = atoi(e);
  const VariantOps* var = default_variant();
  if (u != 0 || w != 0 || gpw != 0) {
    if (u == 0) u = var->lanes;
    if (gpw == 0) gpw = u == var->lanes ? var->kind : 1;   // the default's kind of kernel where it has the lanes asked for
    if (w == 0 && u == var->lanes && gpw == var->kind) w = var->wpe;
    var = find_variant(u, w, gpw);
  }
  if (!var) {
    g_create_error = "monsoon_create: no kernel variant for lanes_per_game=" + std::to_string(u) + " waves_per_simd=" + std::to_string(w) +
                     " in this build (monsoon_amd/csrc/variants.def)";
    return MONSOON_ERR_ARG;
  }
  monsoon* h = new monsoon();
  memset(&h->b, 0, sizeof(h->b));
  h->cfg = *cfg;
  h->var = var;
  // cfg.stack_bytes is ignored: the rules core keeps its own work stack (LDS + b.wk_ovf) and the kernels need no
  // per-lane stack in scratch memory
  h->device = cfg->device;
  memset(h->st_acc, 0, sizeof(h->st_acc));
  memset(h->st_base, 0, sizeof(h->st_base));
  int rc = create_impl(h);
  if (rc != MONSOON_OK) {
    g_create_error = "monsoon_create: " + h->err;
    monsoon_destroy(h);
    return rc;
  }
  *out = h;
  return MONSOON_OK;
}

int monsoon_variant(monsoon_t* h, int32_t* lanes_per_game, int32_t* waves_per_simd) {
  if (!h) return MONSOON_ERR_ARG;
  if (lanes_per_game) *lanes_per_game = h->var->lanes;
  if (waves_per_simd) *waves_per_simd = h->var->wpe;
  return MONSOON_OK;
}

// Make the handle's device current.
static hipError_t bind_device(monsoon_t* h) { return hipSetDevice(h->device); }

static int check_ready(monsoon_t* h) {
  if (!h) return MONSOON_ERR_ARG;
  if (h->n <= 0) {
    h->err = "no games loaded: call monsoon_reset first";
    return MONSOON_ERR_STATE;
  }
  hipError_t e = bind_device(h);
  if (e != hipSuccess) {
    h->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return MONSOON_ERR_DEVICE;
  }
  return MONSOON_OK;
}

static int launch_reset(monsoon_t* h, int n) {
  hipLaunchKernelGGL(k_seed, dim3(n), dim3(64), 0, h->stream, h->b, 0, n, h->d_seeds);
  hipLaunchKernelGGL(k_init, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, h->d_decks, h->d_factions, h->d_seeds);
  HIP_TRY(h, hipGetLastError());
  return MONSOON_OK;
}

static int fold_stats(monsoon_t* h);

static int check_decks(monsoon_t* h, const uint8_t* decks, size_t n_cards, const char* who) {
  for (size_t i = 0; i < n_cards; i++) {
    if (decks[i] >= NUM_CARDS || card_unsupported(decks[i])) {
      h->err = std::string(who) + ": card not supported by this build: " + (decks[i] < NUM_CARDS ? kCardIds[decks[i]] : "index out of range");
      return MONSOON_ERR_ARG;
    }
  }
  return MONSOON_OK;
}

int monsoon_reset(monsoon_t* h, int32_t n, const uint32_t* seeds, const uint8_t* decks, const uint8_t* factions) {
  if (!h || !seeds || !decks || n <= 0 || n > h->cfg.max_games) {
    if (h) h->err = "monsoon_reset: bad argument";
    return MONSOON_ERR_ARG;
  }
  int rc = check_decks(h, decks, (size_t)n * 24, "monsoon_reset");
  if (rc) return rc;
  HIP_TRY(h, bind_device(h));
  rc = fold_stats(h);   // statistics live in the per-game rows that are about to be cleared
  if (rc) return rc;
  HIP_TRY(h, hipMemcpyAsync(h->d_seeds, seeds, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_decks, decks, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
  if (factions)
    HIP_TRY(h, hipMemcpyAsync(h->d_factions, factions, (size_t)n * 2, hipMemcpyHostToDevice, h->stream));
  else
    HIP_TRY(h, hipMemsetAsync(h->d_factions, 0, (size_t)n * 2, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->b.meta, 0, (size_t)n * sizeof(GameMeta), h->stream));
  rc = launch_reset(h, n);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n = n;
  return MONSOON_OK;
}

int monsoon_draw_decks(monsoon_t* h, const uint32_t* seeds, int32_t n, const uint8_t* pool, int32_t pool_n, uint8_t* out_pairs) {
  if (!h || !seeds || !pool || !out_pairs || n <= 0 || pool_n < 12 || pool_n > 128) {
    if (h) h->err = "monsoon_draw_decks: bad argument (12 <= pool_n <= 128)";
    return MONSOON_ERR_ARG;
  }
  for (int i = 0; i < pool_n; i++)
    if (pool[i] >= NUM_CARDS) {
      h->err = "monsoon_draw_decks: pool entry " + std::to_string(i) + " is not a card index";
      return MONSOON_ERR_ARG;
    }
  HIP_TRY(h, bind_device(h));
  uint32_t* d_seeds = nullptr;
  uint8_t *d_pool = nullptr, *d_out = nullptr;
  int* d_over = nullptr;
  auto done = [&](int rc) {
    hipFree(d_seeds);
    hipFree(d_pool);
    hipFree(d_out);
    hipFree(d_over);
    return rc;
  };
#define DD_TRY(call)                                                  \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);     \
      return done(MONSOON_ERR_DEVICE);                                \
    }                                                                 \
  } while (0)
  DD_TRY(hipMalloc(&d_seeds, (size_t)n * 4));
  DD_TRY(hipMalloc(&d_pool, 128));
  DD_TRY(hipMalloc(&d_out, (size_t)n * 24));
  DD_TRY(hipMalloc(&d_over, 4));
  DD_TRY(hipMemcpyAsync(d_seeds, seeds, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  DD_TRY(hipMemcpyAsync(d_pool, pool, (size_t)pool_n, hipMemcpyHostToDevice, h->stream));
  DD_TRY(hipMemsetAsync(d_over, 0, 4, h->stream));
  hipLaunchKernelGGL(k_draw_decks, dim3(n), dim3(64), 0, h->stream, n, d_seeds, d_pool, pool_n, d_out, d_over);
  DD_TRY(hipGetLastError());
  int over = 0;
  DD_TRY(hipMemcpyAsync(out_pairs, d_out, (size_t)n * 24, hipMemcpyDeviceToHost, h->stream));
  DD_TRY(hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, h->stream));
  DD_TRY(hipStreamSynchronize(h->stream));
#undef DD_TRY
  if (over) {
    h->err = "monsoon_draw_decks: " + std::to_string(over) + " seed(s) needed more than 1248 outputs of their stream";
    return done(MONSOON_ERR_STATE);
  }
  return done(MONSOON_OK);
}

int monsoon_ga_offspring(monsoon_t* h, monsoon_np_state* st, const double* parents_w, const double* parents_s, int32_t mu, int32_t dim,
                         int32_t lambda, double tau, double tau_prime, double min_sigma, double* out_w, double* out_s, int32_t* out_parent,
                         int64_t* out_tries) {
  if (!h || !st || !parents_w || !parents_s || !out_w || !out_s || mu <= 0 || lambda <= 0 || dim <= 0 || dim > 16 || st->pos < 0 || st->pos > MT_N) {
    if (h) h->err = "monsoon_ga_offspring: bad argument (1 <= dim <= 16, 0 <= state position <= 624)";
    return MONSOON_ERR_ARG;
  }
  HIP_TRY(h, bind_device(h));
  const size_t pb = (size_t)mu * dim * 8, ob = (size_t)lambda * dim * 8;
  uint8_t* d = nullptr;   // one allocation: key | pos, has_gauss | gauss | parents w, s | out w, s | parent | tries
  const size_t o_key = 0, o_pg = MT_N * 4, o_g = o_pg + 8, o_pw = o_g + 8, o_ps = o_pw + pb, o_ow = o_ps + pb, o_os = o_ow + ob, o_par = o_os + ob,
               o_tr = (o_par + (size_t)lambda * 4 + 7) & ~(size_t)7, total = o_tr + (size_t)lambda * 8;
  HIP_TRY(h, hipMalloc(&d, total));
  auto done = [&](int rc) {
    hipFree(d);
    return rc;
  };
#define GA_TRY(call)                                                  \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);     \
      return done(MONSOON_ERR_DEVICE);                                \
    }                                                                 \
  } while (0)
  int pg[2] = {st->pos, st->has_gauss};
  GA_TRY(hipMemcpyAsync(d + o_key, st->key, MT_N * 4, hipMemcpyHostToDevice, h->stream));
  GA_TRY(hipMemcpyAsync(d + o_pg, pg, 8, hipMemcpyHostToDevice, h->stream));
  GA_TRY(hipMemcpyAsync(d + o_g, &st->gauss, 8, hipMemcpyHostToDevice, h->stream));
  GA_TRY(hipMemcpyAsync(d + o_pw, parents_w, pb, hipMemcpyHostToDevice, h->stream));
  GA_TRY(hipMemcpyAsync(d + o_ps, parents_s, pb, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_ga_offspring, dim3(1), dim3(64), 0, h->stream, (uint32_t*)(d + o_key), (int*)(d + o_pg), (double*)(d + o_g),
                     (const double*)(d + o_pw), (const double*)(d + o_ps), mu, dim, lambda, tau, tau_prime, min_sigma, (double*)(d + o_ow),
                     (double*)(d + o_os), (int*)(d + o_par), (long long*)(d + o_tr));
  GA_TRY(hipGetLastError());
  GA_TRY(hipMemcpyAsync(st->key, d + o_key, MT_N * 4, hipMemcpyDeviceToHost, h->stream));
  GA_TRY(hipMemcpyAsync(pg, d + o_pg, 8, hipMemcpyDeviceToHost, h->stream));
  GA_TRY(hipMemcpyAsync(&st->gauss, d + o_g, 8, hipMemcpyDeviceToHost, h->stream));
  GA_TRY(hipMemcpyAsync(out_w, d + o_ow, ob, hipMemcpyDeviceToHost, h->stream));
  GA_TRY(hipMemcpyAsync(out_s, d + o_os, ob, hipMemcpyDeviceToHost, h->stream));
  if (out_parent) GA_TRY(hipMemcpyAsync(out_parent, d + o_par, (size_t)lambda * 4, hipMemcpyDeviceToHost, h->stream));
  if (out_tries) GA_TRY(hipMemcpyAsync(out_tries, d + o_tr, (size_t)lambda * 8, hipMemcpyDeviceToHost, h->stream));
  GA_TRY(hipStreamSynchronize(h->stream));
  st->pos = pg[0];
  st->has_gauss = pg[1];
  return done(MONSOON_OK);
}

int monsoon_ga_select(monsoon_t* h, const double* fitness, int32_t n, int32_t* out_order) {
  if (!h || !fitness || !out_order || n <= 0) {
    if (h) h->err = "monsoon_ga_select: bad argument";
    return MONSOON_ERR_ARG;
  }
  for (int i = 0; i < n; i++)
    if (fitness[i] != fitness[i]) {
      h->err = "monsoon_ga_select: NaN fitness (Python's sort order would be undefined)";
      return MONSOON_ERR_ARG;
    }
  HIP_TRY(h, bind_device(h));
  uint8_t* d = nullptr;
  HIP_TRY(h, hipMalloc(&d, (size_t)n * 12));
  auto done = [&](int rc) {
    hipFree(d);
    return rc;
  };
  GA_TRY(hipMemcpyAsync(d, fitness, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_ga_select, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const double*)d, n, (int*)(d + (size_t)n * 8));
  GA_TRY(hipGetLastError());
  GA_TRY(hipMemcpyAsync(out_order, d + (size_t)n * 8, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  GA_TRY(hipStreamSynchronize(h->stream));
#undef GA_TRY
  return done(MONSOON_OK);
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
  // Every action is checked against its game's legal set BEFORE any game is stepped: an illegal entry refuses the
  // whole call and leaves every game untouched.
  {
    std::vector<uint64_t> masks((size_t)n * 3);
    rc = monsoon_legal_mask(h, masks.data());
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
      int a = actions[i];
      if (a == 255 || a == 155) continue;   // skip / PASS (always accepted, see k_step)
      if (a >= MONSOON_NUM_ACTIONS || !((masks[(size_t)i * 3 + (a >> 6)] >> (a & 63)) & 1)) {
        h->err = "monsoon_step: illegal action " + std::to_string(a) + " for game " + std::to_string(i);
        return MONSOON_ERR_ARG;
      }
    }
  }
  uint8_t* d = h->d_bytes;   // [actions | reward | done | fault | illegal] x n
  HIP_TRY(h, hipMemcpyAsync(d, actions, n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_step, dim3((n + API_LANES - 1) / API_LANES), dim3(64), API_LDS_BYTES, h->stream, h->b, n, d, (int8_t*)(d + n), d + 2 * (size_t)n,
                     d + 3 * (size_t)n, d + 4 * (size_t)n);
  HIP_TRY(h, hipGetLastError());
  std::vector<uint8_t> host(4 * (size_t)n);
  HIP_TRY(h, hipMemcpyAsync(host.data(), d + n, 4 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
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
  hipLaunchKernelGGL(k_status, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->b, n, h->d_i32);
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

// copy.deepcopy(game) across the boundary: the complete device state of one game (record, meta row, stream) as an
// opaque blob that monsoon_state_load puts back into any slot of any handle of the same build.
int32_t monsoon_state_blob_bytes(void) { return BLOB_BYTES; }

int monsoon_state_save(monsoon_t* h, int32_t idx, uint8_t* buf, int32_t buf_bytes) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!buf || buf_bytes < BLOB_BYTES || idx < 0 || idx >= h->n) {
    h->err = "monsoon_state_save: bad argument (buffer must hold monsoon_state_blob_bytes())";
    return MONSOON_ERR_ARG;
  }
  static_assert(BLOB_BYTES <= 16384, "blob staging buffer");
  hipLaunchKernelGGL(k_blob, dim3(1), dim3(256), 0, h->stream, h->b, idx, (uint32_t*)h->d_bytes, 0);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(buf, h->d_bytes, BLOB_BYTES, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

// idx may be any slot below max_games; loading into slot h->n appends a game to the loaded batch.
int monsoon_state_load(monsoon_t* h, int32_t idx, const uint8_t* buf, int32_t buf_bytes) {
  if (!h || !buf) return MONSOON_ERR_ARG;
  uint32_t head[2];
  if (buf_bytes >= 8) memcpy(head, buf, 8);
  if (buf_bytes < BLOB_BYTES || head[0] != BLOB_MAGIC || head[1] != (uint32_t)STATE_BYTES || idx < 0 || idx > h->n || idx >= h->cfg.max_games) {
    h->err = "monsoon_state_load: not a state blob of this build, or slot out of range";
    return MONSOON_ERR_ARG;
  }
  // The blob carries the game's bookkeeping row as it was in the handle that saved it.  What refers to THAT handle is
  // checked here: a stream cursor outside its two resident blocks is refused; the players' weight rows are kept only if
  // this handle's weight table has them (else both become row 0 -- assign players again before deciding).  The schedule
  // index travels along untouched: only monsoon_rollout reads it, after it has reset every game and set it anew.
  GameMeta gm;
  memcpy(&gm, buf + 8, sizeof(gm));
  if ((gm.rng & 0xffffu) >= (uint32_t)(2 * MT_N) || (gm.rng >> 17) != 0) {
    h->err = "monsoon_state_load: the blob's stream cursor is out of range";
    return MONSOON_ERR_ARG;
  }
  if (gm.p1 < 0 || gm.p2 < 0 || (h->b.weights && (gm.p1 >= h->n_individuals || gm.p2 >= h->n_individuals))) gm.p1 = gm.p2 = 0;
  std::vector<uint8_t> staged(buf, buf + BLOB_BYTES);
  memcpy(staged.data() + 8, &gm, sizeof(gm));
  HIP_TRY(h, bind_device(h));
  HIP_TRY(h, hipMemcpy(h->d_bytes, staged.data(), BLOB_BYTES, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_blob, dim3(1), dim3(256), 0, h->stream, h->b, idx, (uint32_t*)h->d_bytes, 1);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (idx == h->n) h->n = idx + 1;
  return MONSOON_OK;
}

// Diagnostics for the scenario tests (tests/scenario_lib.py): build game idx from a state stream / make one engine call.
static int debug_call(monsoon_t* h, int32_t idx, int build, uint32_t seed, uint32_t stream_pos, const int32_t* stream, int32_t n_stream,
                      int32_t* fault, int32_t* log, int32_t log_cap, int32_t* n_log) {
  if (!h || !stream || n_stream <= 0 || n_stream > 3000 || idx < 0 || idx >= h->cfg.max_games || idx > h->n) {
    if (h) h->err = "monsoon_debug_*: bad argument";
    return MONSOON_ERR_ARG;
  }
  HIP_TRY(h, bind_device(h));
  static_assert(3000 * 4 + (2 + 2 * DBG_TRACE_CAP) * 4 <= 16384, "staging buffer");
  int32_t* d_stream = (int32_t*)h->d_bytes;
  int32_t* d_out = d_stream + 3000;
  HIP_TRY(h, hipMemcpyAsync(d_stream, stream, (size_t)n_stream * 4, hipMemcpyHostToDevice, h->stream));
  if (build) {
    HIP_TRY(h, hipMemcpyAsync(h->d_seeds + idx, &seed, 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, h->stream, h->b, idx, 1, h->d_seeds + idx);
  }
  hipLaunchKernelGGL(k_debug, dim3(1), dim3(64), DBG_LDS_BYTES, h->stream, h->b, idx, build, seed, stream_pos, d_stream, d_out);
  HIP_TRY(h, hipGetLastError());
  std::vector<int32_t> out(2 + 2 * DBG_TRACE_CAP);
  HIP_TRY(h, hipMemcpyAsync(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (fault) *fault = out[0];
  int n = out[1] < log_cap ? out[1] : log_cap;
  if (n_log) *n_log = n;
  if (log && n > 0) memcpy(log, out.data() + 2, (size_t)n * 8);
  if (build && idx == h->n) h->n = idx + 1;
  return MONSOON_OK;
}
int monsoon_debug_build(monsoon_t* h, int32_t idx, uint32_t seed, uint32_t stream_pos, const int32_t* state, int32_t n_state, int32_t* fault) {
  return debug_call(h, idx, 1, seed, stream_pos, state, n_state, fault, nullptr, 0, nullptr);
}
int monsoon_debug_op(monsoon_t* h, int32_t idx, const int32_t* op, int32_t n_op, int32_t* fault, int32_t* log, int32_t log_cap, int32_t* n_log) {
  if (h && idx >= h->n) {
    h->err = "monsoon_debug_op: no such game";
    return MONSOON_ERR_ARG;
  }
  return debug_call(h, idx, 0, 0, 0, op, n_op, fault, log, log_cap, n_log);
}

// kind 0: n raw u32 draws of RandomState(seed); 1: n random() doubles; 2: randint(0, in[i]) for n int32 bounds; 3: n x
// shuffle(list(range(12))) on one stream (out int32[n][12]); 4: n scores of in = double[n][30] {weights, before, after}.
// Uses (and re-seeds) the stream buffers of game slot 0.
int monsoon_debug_kat(monsoon_t* h, int32_t kind, uint32_t seed, int32_t n, const void* in, void* out) {
  if (!h || !out || n <= 0 || n > 4096 || kind < 0 || kind > 4 || ((kind == 2 || kind == 4) && !in)) {
    if (h) h->err = "monsoon_debug_kat: bad argument";
    return MONSOON_ERR_ARG;
  }
  HIP_TRY(h, bind_device(h));
  const size_t in_bytes = kind == 2 ? (size_t)n * 4 : (kind == 4 ? (size_t)n * 240 : 0);
  const size_t out_bytes = kind == 3 ? (size_t)n * 48 : (kind == 1 || kind == 4 ? (size_t)n * 8 : (size_t)n * 4);
  void *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(h, hipMalloc(&d_out, out_bytes));
  if (in_bytes) {
    hipError_t e = hipMalloc(&d_in, in_bytes);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      hipFree(d_out);
      if (d_in) hipFree(d_in);
      h->err = std::string("monsoon_debug_kat: ") + hipGetErrorString(e);
      return MONSOON_ERR_DEVICE;
    }
  }
  hipMemcpyAsync(h->d_seeds, &seed, 4, hipMemcpyHostToDevice, h->stream);
  hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, h->stream, h->b, 0, 1, h->d_seeds);
  hipLaunchKernelGGL(k_kat, dim3(1), dim3(64), DBG_LDS_BYTES, h->stream, h->b, kind, n, (const void*)d_in, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(d_out);
  if (d_in) hipFree(d_in);
  if (e != hipSuccess) {
    h->err = std::string("monsoon_debug_kat: ") + hipGetErrorString(e);
    return MONSOON_ERR_DEVICE;
  }
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
  HIP_TRY(h, bind_device(h));
  if (n_individuals > h->weights_cap) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->b.weights) HIP_TRY(h, hipFree(h->b.weights));
    h->b.weights = nullptr;
    h->weights_cap = 0;
    h->n_individuals = 0;
    HIP_TRY(h, hipMalloc(&h->b.weights, (size_t)n_individuals * 80));
    h->weights_cap = n_individuals;
  }
  HIP_TRY(h, hipMemcpyAsync(h->b.weights, weights, (size_t)n_individuals * 80, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n_individuals = n_individuals;   // rows beyond the uploaded count are stale: assign_players refuses them
  return MONSOON_OK;
}

static int assign_players(monsoon_t* h, const int32_t* p1, const int32_t* p2, int match_base) {
  int n = h->n;
  for (int i = 0; i < n; i++)
    if (p1[i] < 0 || p2[i] < 0 || p1[i] >= h->n_individuals || p2[i] >= h->n_individuals) {
      h->err = "monsoon_assign_players: index outside the uploaded weight table";
      return MONSOON_ERR_ARG;
    }
  HIP_TRY(h, hipMemcpyAsync(h->d_p1, p1, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_p2, p2, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_assign, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, h->d_p1, h->d_p2, match_base);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}

int monsoon_assign_players(monsoon_t* h, const int32_t* p1, const int32_t* p2) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!p1 || !p2) return MONSOON_ERR_ARG;
  return assign_players(h, p1, p2, 0);
}

// a reusable pair of timing events (created once per handle, recycled by drain_timing)
static int timing_begin(monsoon_t* h, size_t* slot) {
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(h, hipEventCreate(&e0));
    hipError_t e = hipEventCreate(&e1);
    if (e != hipSuccess) {
      hipEventDestroy(e0);
      h->err = std::string("hipEventCreate: ") + hipGetErrorString(e);
      return MONSOON_ERR_DEVICE;
    }
    h->ev_pool.emplace_back(e0, e1);
  }
  *slot = h->ev_used++;
  HIP_TRY(h, hipEventRecord(h->ev_pool[*slot].first, h->stream));
  return MONSOON_OK;
}

static int drain_timing(monsoon_t* h) {
  for (size_t i = 0; i < h->ev_used; i++) {
    auto& pr = h->ev_pool[i];
    HIP_TRY(h, hipEventSynchronize(pr.second));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, pr.first, pr.second));
    h->kernel_ms += ms;
    h->kernel_launches++;
  }
  h->ev_used = 0;
  return MONSOON_OK;
}

static const int g_persistent = getenv("MONSOON_PERSIST") ? atoi(getenv("MONSOON_PERSIST")) : 1;
static const int g_lds_pad = getenv("MONSOON_LDS_PAD") ? atoi(getenv("MONSOON_LDS_PAD")) : 0;   // occupancy experiments only

// `rounds` decisions of every loaded game in ONE launch (1 = a decision round; max_turns + 1 = whole games: the extra
// round turns "still running at the cap" into a result).
static int launch_play(monsoon_t* h, int n, int max_turns, int rounds, int write_scores, bool timed) {
  size_t slot = 0;
  if (timed) {
    int rc = timing_begin(h, &slot);
    if (rc) return rc;
  }
  const VariantOps* v = h->var;
  const int lds = v->lds_bytes + g_lds_pad;
  if (!h->grid_waves) {   // resident wavefronts of the handle's kernel variant (queried once)
    int per_cu = 0;
    hipDeviceProp_t prop;
    HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));
    HIP_TRY(h, v->occupancy(&per_cu, lds));
    h->grid_waves = per_cu > 0 ? per_cu * prop.multiProcessorCount : 4096;
  }
  // one decision per game and launch: two waves per slot measured best; many decisions per game: exactly the resident waves
  int grid = rounds > 1 ? h->grid_waves : 2 * h->grid_waves;
  if (const char* e = getenv("MONSOON_GRID")) grid = atoi(e);
  // the persistent form needs a wavefront for every one of its POP_PARTS ranges
  const int pers = (g_persistent && (long long)grid * v->games < n && grid >= POP_PARTS) ? 1 : 0;
  if (!pers) grid = (n + v->games - 1) / v->games;   // a wavefront per v->games games
  if ((size_t)grid * v->lanes * v->games > h->ovf_lanes) {   // work-stack overflow blocks for every workgroup of this grid
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(h->b.wk_ovf));
    h->b.wk_ovf = nullptr;
    h->ovf_lanes = (size_t)grid * v->lanes * v->games;
    HIP_TRY(h, hipMalloc(&h->b.wk_ovf, h->ovf_lanes * OVF_WORDS * 4));
  }
  v->play(grid, lds, h->stream, h->b, n, max_turns, rounds, write_scores, pers, h->parity);
  // Only a persistent launch consumes its counter set and clears the other one: a non-persistent launch in between
  // must leave the parity alone, or the next persistent launch would start from the stale counts of the one before.
  if (pers) h->parity ^= 1;
  HIP_TRY(h, hipGetLastError());
  if (timed) HIP_TRY(h, hipEventRecord(h->ev_pool[slot].second, h->stream));
  return MONSOON_OK;
}

int monsoon_decide_round_dev(monsoon_t* h) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!h->b.weights) {
    h->err = "monsoon_decide_round_dev: upload weights and assign players first";
    return MONSOON_ERR_STATE;
  }
  return launch_play(h, h->n, 0x7fff, 1, 0, true);
}

// `rounds` decisions of every loaded game in one launch: a game's record stays in LDS from its first to its last decision
// of the call (asynchronous on the handle's stream, like monsoon_decide_round_dev = rounds 1).
int monsoon_play_rounds_dev(monsoon_t* h, int32_t rounds) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (rounds <= 0 || !h->b.weights) {
    h->err = "monsoon_play_rounds_dev: rounds must be positive; upload weights and assign players first";
    return rounds <= 0 ? MONSOON_ERR_ARG : MONSOON_ERR_STATE;
  }
  return launch_play(h, h->n, 0x7fff, rounds, 0, true);
}

int monsoon_sync(monsoon_t* h) {
  if (!h) return MONSOON_ERR_ARG;
  HIP_TRY(h, bind_device(h));
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
  rc = assign_players(h, p1.data(), p2.data(), 0);
  if (rc) return rc;
  if (out_scores) {
    if (!h->b.scores) HIP_TRY(h, hipMalloc(&h->b.scores, (size_t)h->cfg.max_games * MONSOON_NUM_ACTIONS * 8));
    size_t cnt = (size_t)n * MONSOON_NUM_ACTIONS;
    hipLaunchKernelGGL(k_clear_scores, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->b.scores, cnt);
  }
  rc = launch_play(h, n, 0x7fff, 1, out_scores ? 1 : 0, false);
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
  if (!h) return MONSOON_ERR_ARG;
  if (!weights || !matches || !deck_pairs || !out_counts || n_matches <= 0 || n_individuals <= 0 || n_decks <= 0 || max_turns <= 0 ||
      max_turns > 30000) {
    h->err = "monsoon_rollout: bad argument";
    return MONSOON_ERR_ARG;
  }
  // The whole schedule and every deck are checked before anything is loaded or launched.
  for (int i = 0; i < n_matches; i++) {
    const monsoon_match& mm = matches[i];
    if (mm.p1 < 0 || mm.p1 >= n_individuals || mm.p2 < 0 || mm.p2 >= n_individuals || mm.deck >= (uint32_t)n_decks) {
      h->err = "monsoon_rollout: schedule entry " + std::to_string(i) + " out of range";
      return MONSOON_ERR_ARG;
    }
  }
  int rc = check_decks(h, deck_pairs, (size_t)n_decks * 24, "monsoon_rollout");
  if (rc) return rc;
  rc = monsoon_upload_weights(h, weights, n_individuals);
  if (rc) return rc;
  // result buffers live in the handle: grown when a larger schedule arrives, never freed per call
  if ((size_t)n_individuals > h->counts_cap) {
    if (h->d_counts) HIP_TRY(h, hipFree(h->d_counts));
    h->d_counts = nullptr;
    h->counts_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_counts, (size_t)n_individuals * 12));
    h->counts_cap = (size_t)n_individuals;
  }
  if ((size_t)n_matches > h->matches_cap) {
    if (h->d_results) HIP_TRY(h, hipFree(h->d_results));
    if (h->d_steps) HIP_TRY(h, hipFree(h->d_steps));
    h->d_results = nullptr;
    h->d_steps = nullptr;
    h->matches_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_results, 2 * (size_t)n_matches));
    HIP_TRY(h, hipMalloc(&h->d_steps, (size_t)n_matches * 4));
    h->matches_cap = (size_t)n_matches;
  }
  HIP_TRY(h, hipMemsetAsync(h->d_counts, 0, (size_t)n_individuals * 12, h->stream));
  h->rollout_matches = 0;
  uint8_t* d_faults = (uint8_t*)h->d_results + h->matches_cap;
  int cap = h->cfg.max_games;
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
      seeds[i] = mm.seed;
      memcpy(&decks[(size_t)i * 24], deck_pairs + (size_t)mm.deck * 24, 24);
      p1[i] = mm.p1;
      p2[i] = mm.p2;
    }
    rc = monsoon_reset(h, n, seeds.data(), decks.data(), nullptr);
    if (!rc) rc = assign_players(h, p1.data(), p2.data(), base);
    if (rc) return rc;
    // one launch plays the whole batch to the end
    rc = launch_play(h, n, max_turns, max_turns + 1, 0, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_collect, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->b, n, h->d_counts, h->d_results, h->d_steps, d_faults);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    rc = drain_timing(h);
    if (rc) return rc;
  }
  std::vector<int32_t> counts((size_t)n_individuals * 3);
  HIP_TRY(h, hipMemcpy(counts.data(), h->d_counts, counts.size() * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < counts.size(); i++) out_counts[i] += counts[i];
  if (out_results) HIP_TRY(h, hipMemcpy(out_results, h->d_results, (size_t)n_matches, hipMemcpyDeviceToHost));
  if (out_steps) HIP_TRY(h, hipMemcpy(out_steps, h->d_steps, (size_t)n_matches * 4, hipMemcpyDeviceToHost));
  h->rollout_matches = (size_t)n_matches;
  return MONSOON_OK;
}

int monsoon_rollout_faults(monsoon_t* h, uint8_t* out, int32_t n_matches) {
  if (!h || !out) return MONSOON_ERR_ARG;
  if (n_matches <= 0 || (size_t)n_matches != h->rollout_matches) {
    h->err = "monsoon_rollout_faults: n_matches is not the size of the last completed monsoon_rollout";
    return MONSOON_ERR_ARG;
  }
  HIP_TRY(h, bind_device(h));
  HIP_TRY(h, hipMemcpy(out, (uint8_t*)h->d_results + h->matches_cap, (size_t)n_matches, hipMemcpyDeviceToHost));
  return MONSOON_OK;
}

// current statistics of the loaded games (see k_stats)
static int reduce_stats(monsoon_t* h, unsigned long long cur[ST_N]) {
  for (int i = 0; i < ST_N; i++) cur[i] = 0;
  if (h->n <= 0) return MONSOON_OK;
  HIP_TRY(h, hipMemsetAsync(h->b.stats, 0, ST_N * sizeof(unsigned long long), h->stream));
  int blocks = (h->n + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(k_stats, dim3(blocks), dim3(256), 0, h->stream, h->b, h->n, h->b.stats);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(cur, h->b.stats, ST_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MONSOON_OK;
}
static int total_stats(monsoon_t* h, unsigned long long tot[ST_N]) {
  unsigned long long cur[ST_N];
  int rc = reduce_stats(h, cur);
  if (rc) return rc;
  for (int i = 0; i < ST_N; i++) tot[i] = h->st_acc[i] + cur[i] - h->st_base[i];
  return MONSOON_OK;
}
// the loaded games are about to be replaced: keep what they contributed
static int fold_stats(monsoon_t* h) {
  unsigned long long tot[ST_N];
  int rc = total_stats(h, tot);
  if (rc) return rc;
  for (int i = 0; i < ST_N; i++) {
    h->st_acc[i] = tot[i];
    h->st_base[i] = 0;
  }
  return MONSOON_OK;
}

int monsoon_get_stats(monsoon_t* h, monsoon_stats* out) {
  if (!h || !out) return MONSOON_ERR_ARG;
  HIP_TRY(h, bind_device(h));
  unsigned long long s[ST_N];
  int rc = total_stats(h, s);
  if (rc) return rc;
  out->lookahead_steps = s[ST_LOOKAHEAD];
  out->decisions = s[ST_DECISIONS];
  out->games_finished = s[ST_FINISHED];
  out->faults = s[ST_FAULTS];
  out->capacity_faults = s[ST_CAPFAULTS];
  out->lookahead_capacity_faults = s[ST_LACAPFAULTS];
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
  HIP_TRY(h, bind_device(h));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  drain_timing(h);
  {
    unsigned long long cur[ST_N];
    int rc = reduce_stats(h, cur);
    if (rc) return rc;
    for (int i = 0; i < ST_N; i++) {
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
  HIP_TRY(h, bind_device(h));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  int rc = drain_timing(h);
  if (rc) return rc;
  if (total_ms) *total_ms = h->kernel_ms;
  if (launches) *launches = h->kernel_launches;
  return MONSOON_OK;
}

void* monsoon_stream(monsoon_t* h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"
