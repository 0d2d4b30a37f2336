// kernels_reg.h -- k_play_reg<U, W>: k_play with the game's current record and the best successor so far in REGISTERS.
//
// A record is SG granules of 16 bytes: lane l of the wavefront keeps granules l, l + 64, ... of the current record (v_par)
// and of the best successor found so far in this decision (v_best) -- one granule = four VGPRs each on the standard record.  What k_play keeps in LDS besides the U
// candidate columns -- a copy of the current record -- is gone (752 bytes per wavefront: 22 instead of 20 wavefronts fit a
// CU's LDS), and so are the round trips through HBM that parked the best successor of a decision with more than U legal
// actions: cloning is U 16-byte LDS writes per lane straight from registers, a new best is one 16-byte read per lane, the
// commit is a register move.  The functions that read the current record (end of game, legal mask, features, stream
// cursor) find its image in candidate column 0, written there from the registers before they run (Col0Mem, state.h).
// Everything a game computes is what k_play computes for it.
#pragma once
#include "kernels.h"

namespace msbk {

template <int U>
struct RegLds {
  static constexpr int PRIV = LDS_ORIGIN;
  static constexpr int PRIV_BYTES = SG * U * 16 > MT_N * 4 ? SG * U * 16 : ((MT_N * 4 + 15) & ~15);   // doubles as the twist buffer
  static constexpr int WF = PRIV + PRIV_BYTES;       // 10 weights + 10 "before" features + 10 features of the best successor (f64)
  static constexpr int SKB = WF + 240;               // work stacks of the U candidate lanes / their ten "after" features each
  static constexpr int CF = SKB;
  static constexpr int TOTAL = SKB + U * SKW * 4;
};

template <int U>
__device__ MSB_INL void play_game_reg(const DevBuffers& b, const int g, const int lane, int max_turns, int rounds, int write_scores) {
  typedef RegLds<U> L;
  typedef Engine<Col0Mem<U, L::PRIV>> ParEngine;
  typedef Engine<LaneMem<U, L::PRIV, L::SKB, SKW>> CandEngine;
  constexpr int GPL = (SG + 63) / 64;   // granules of a record per lane
  GameMeta meta = b.meta[g];
  if (meta.result != -2) {
    if (lane == 0) {
      b.meta[g].last_action = 255;
      if (b.best) b.best[g] = NAN;
    }
    return;
  }
  MSB_AS_LDS u32x4* priv = (MSB_AS_LDS u32x4*)(uintptr_t)L::PRIV;
  MSB_AS_LDS double* wf = (MSB_AS_LDS double*)(uintptr_t)L::WF;
  MSB_AS_LDS double* cf = (MSB_AS_LDS double*)(uintptr_t)(L::CF + (lane < U ? lane : 0) * 80);
  u32x4* grec = (u32x4*)(b.state + (size_t)g * SW);
  u32x4 v_par[GPL], v_best[GPL];
#define MSB_EACH_GRANULE(body_)                   \
  _Pragma("unroll") for (int j_ = 0; j_ < GPL; j_++) { \
    const int gr_ = lane + 64 * j_;               \
    if (gr_ < SG) { body_; }                      \
  }
  _Pragma("unroll") for (int j_ = 0; j_ < GPL; j_++) v_par[j_] = u32x4{0u, 0u, 0u, 0u};
  MSB_EACH_GRANULE(v_par[j_] = grec[gr_])   // coalesced 16-B-per-lane loads, straight into registers
  _Pragma("unroll") for (int j_ = 0; j_ < GPL; j_++) v_best[j_] = v_par[j_];
  ParEngine pe;
  CandEngine ce;
  // the image of the current record in column 0; with the stream window attached it is also what the clones start from
  MSB_EACH_GRANULE(priv[gr_ * U] = v_par[j_])
  __syncthreads();
  if (lane == 0) attach_rng(pe, b, g, meta.rng);
  __syncthreads();
  MSB_EACH_GRANULE(v_par[j_] = priv[gr_ * U])
  bool have_before = false;
  double last_score = NAN;
  int played = 0;

  for (int round = 0; round < rounds; round++) {   // (column 0 == v_par here)
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
    const msb_u64x4 lm = pe.legal_mask_v();
    const uint64_t mask[3] = {uni64(lm[0]), uni64(lm[1]), uni64(lm[2])};
    uint64_t rem[3] = {mask[0], mask[1], mask[2]};
    const int n_legal = __popcll(mask[0]) + __popcll(mask[1]) + __popcll(mask[2]);
    const bool before_raises = pe.observation_raises();
    {
      const double* wt = b.weights + (size_t)(pe.local() == 0 ? meta.p1 : meta.p2) * 10;
      if (lane < 10) wf[lane] = wt[lane];
      if (!before_raises && !have_before) {
        if (lane == 0) pe.features(wf + 10);
      }
    }
    __syncthreads();

    constexpr int NONE_A = 1 << 20;
    double run_s = 0.0;
    int run_a = NONE_A;
    int cfault = 0;
    int feat_ok = 0;
    int la_fault = 0;
    for (int base = 0; base < n_legal; base += U) {
      int k = base + lane;
      double s = 0.0;   // except Exception -> 0.0 (evo/heuristic_agent.py:48-51)
      int a = NONE_A;
      int my_fault = 0;
      int my_feat = 0;
      int f = 0;
      bool raises = false;
      {   // copy.deepcopy (stream window included) for the whole pass: lane l writes granule l of every column in use
        const int n_act = n_legal - base < U ? n_legal - base : U;
        __syncthreads();
        for (int col = 0; col < n_act; col++) MSB_EACH_GRANULE(priv[gr_ * U + col] = v_par[j_])
        __syncthreads();
      }
      const bool active = lane < U && k < n_legal;
      if (active) a = nth_set_bit(rem, lane);
      for (int i = 0; i < U; i++) {   // uniform: drop this pass's actions
        if (rem[0]) rem[0] &= rem[0] - 1;
        else if (rem[1]) rem[1] &= rem[1] - 1;
        else rem[2] &= rem[2] - 1;
      }
      if (active) {
        ce.step(a);
        f = ce.fault();
        raises = f == 0 && ce.observation_raises();
      }
      if (active) {
        if (f == 0 && !before_raises && !raises) {
          double fa[10];
          ce.features(fa);
          for (int i = 0; i < 10; i++) cf[i] = fa[i];
          s = CandEngine::action_score_lds(wf, fa);
          my_feat = 1;
        }
        if (write_scores) b.scores[(size_t)g * MONSOON_NUM_ACTIONS + a] = s;
        my_fault = f ? f : (raises ? FAULT_INT_CARD : 0);
      }
      {
        const unsigned long long lfb = __ballot(active && f >= FAULT_CAPACITY);
        if (lfb && !la_fault) la_fault = __builtin_amdgcn_readlane(f, __builtin_ctzll(lfb));
      }
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
        const int wl = __ffsll((long long)bal) - 1;
        cfault = __builtin_amdgcn_readlane(my_fault, wl);
        feat_ok = __builtin_amdgcn_readlane(my_feat, wl);
        if (feat_ok && lane < 10)   // the winner keeps its features: the next decision's "before" side
          wf[20 + lane] = ((MSB_AS_LDS const double*)(uintptr_t)L::CF)[wl * 10 + lane];
        __syncthreads();
        MSB_EACH_GRANULE(v_best[j_] = priv[gr_ * U + wl])   // the best successor so far, into registers
      }
    }
    // commit: adapter = adapter.apply_action(best).  The successor carries its own stream cursor (H_RNGPOS).
    __syncthreads();
    MSB_EACH_GRANULE(priv[gr_ * U] = v_best[j_])
    __syncthreads();
    {
      uint32_t new_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pe.rng_pos());
      int cur = (meta.rng >> 16) & 1;
      if (new_pos >= (uint32_t)MT_N) {
        new_pos -= MT_N;
        wave_refill(b, g, cur, (MSB_AS_LDS uint32_t*)priv, lane);   // the used-up block becomes the new "next" block
        MSB_EACH_GRANULE(priv[gr_ * U] = v_best[j_])               // (the twist buffer is the column area)
        __syncthreads();
        cur ^= 1;
        if (lane == 0) pe.rng_block_advance();
      }
      meta.rng = new_pos | ((uint32_t)cur << 16);
      if (lane == 0) attach_rng(pe, b, g, meta.rng);
      have_before = feat_ok != 0;
      if (have_before && lane < 10) wf[10 + lane] = wf[20 + lane];
      __syncthreads();
      MSB_EACH_GRANULE(v_par[j_] = priv[gr_ * U])
    }
    meta.steps++;
    meta.last_action = (uint8_t)run_a;
    meta.lookahead += (uint32_t)n_legal;
    meta.decided++;
    if (la_fault && !meta.la_fault) meta.la_fault = (uint8_t)la_fault;
    last_score = run_s;
    played++;
    if (cfault) {
      meta.fault = (uint8_t)cfault;
      meta.result = -1;
      break;
    }
  }
  MSB_EACH_GRANULE(grec[gr_] = v_par[j_])
#undef MSB_EACH_GRANULE
  if (lane == 0) {
    b.meta[g] = meta;
    if (b.best) b.best[g] = last_score;
  }
}

template <int U, int WPE>
__global__ void __launch_bounds__(64, WPE) k_play_reg(DevBuffers b, int n, int max_turns, int rounds, int write_scores, int persistent, int parity) {
  const int lane = threadIdx.x;
  lds_init_wtab(b.wk_ovf + (size_t)blockIdx.x * (U * OVF_WORDS));
  int* mine = b.pop + parity * POP_PARTS * POP_STRIDE;
  int* other = b.pop + (parity ^ 1) * POP_PARTS * POP_STRIDE;
  if (persistent && blockIdx.x == 0 && lane < POP_PARTS) other[lane * POP_STRIDE] = 0;
  const int part = blockIdx.x % POP_PARTS, rank = blockIdx.x / POP_PARTS;
  const int waves = ((int)gridDim.x - part + POP_PARTS - 1) / POP_PARTS;
  const int lo = persistent ? (int)((long long)n * part / POP_PARTS) : 0;
  const int hi = persistent ? (int)((long long)n * (part + 1) / POP_PARTS) : n;
  int t = persistent ? lo + rank : (int)blockIdx.x;
  while (t < hi) {
    int nxt = 0x7fffffff;
    if (persistent && lane == 0) nxt = lo + waves + atomicAdd(&mine[part * POP_STRIDE], 1);
    play_game_reg<U>(b, t, lane, max_turns, rounds, write_scores);
    __syncthreads();
    t = __builtin_amdgcn_readfirstlane(nxt);
  }
}

}  // namespace msbk
