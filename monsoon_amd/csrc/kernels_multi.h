// kernels_multi.h -- k_play_multi<U, G, W>: the hot kernel with G games per wavefront.
//
// k_play (kernels.h) gives a wavefront ONE game: U = 8 of its 64 lanes step candidates, and every vector instruction is
// issued for the whole wavefront whatever the number of live lanes.  Here a wavefront has G game SLOTS; lanes
// [k*U, (k+1)*U) are slot k's candidate lanes.  The slots run the same phases in lockstep -- legal mask, clone, look-ahead
// step, features + score, arg-max, commit -- each on its own game, so one instruction stream serves G games: the
// per-game phases that are the same code for every game (legal mask, features, score, arg-max, commit) cost one
// execution for all slots, and in the look-ahead step the slots share whatever their candidates have in common
// (profiles/r02_probe_divergence.txt: 8 games x 8 actions in one wavefront execute 4x, not 8x, the instructions of one
// game's 8 actions).  A game's LDS footprint is what it is in k_play, so a CU holds the same number of GAMES in
// G times fewer wavefronts.
//
// A slot whose game is over (or has played its `rounds` decisions) writes it back and takes the next game of the
// wavefront's range at once: slots are never idle while their range has games.  Per-game values that k_play keeps in
// scalar registers (legal set, running best, meta row) are per-lane here, replicated over the slot's lanes; copies that
// k_play makes with all 64 lanes (staging, clone, parking, commit, stream refill) are still made with all 64, one slot
// after the other or all slots at once.  Everything a game computes is what k_play computes for it: the GPU tests run
// the same comparisons on either kernel.
#pragma once
#include "kernels.h"

namespace msbk {

template <int U, int G>
struct MultiLds {
  static constexpr int T = U * G;                    // candidate lanes of the wavefront
  static constexpr int PRIV = LDS_ORIGIN;
  static constexpr int PRIV_BYTES = SG * T * 16 > MT_N * 4 ? SG * T * 16 : ((MT_N * 4 + 15) & ~15);   // doubles as the twist buffer
  static constexpr int PAR = PRIV + PRIV_BYTES;      // the slots' current records, one after the other
  static constexpr int WF = PAR + G * SG * 16;       // per slot: 10 weights + 10 "before" features + 10 features of the best successor
  static constexpr int SKB = WF + G * 240;           // work stacks of the T candidate lanes ...
  static constexpr int CF = SKB;                     // ... and, once a pass has stepped, their ten "after" features each
  static constexpr int TOTAL = SKB + T * SKW * 4;
};

template <int U, int G, int WPE>
__global__ void __launch_bounds__(64, WPE) k_play_multi(DevBuffers b, int n, int max_turns, int rounds, int write_scores, int persistent, int parity) {
  typedef MultiLds<U, G> L;
  constexpr int T = L::T;
  static_assert(64 % T == 0 && (U & (U - 1)) == 0, "the slots' lanes tile the wavefront");
  typedef Engine<GroupMem<L::PAR, SG * 16, U>> ParEngine;
  typedef Engine<LaneMem<T, L::PRIV, L::SKB, SKW>> CandEngine;
  constexpr int NONE_G = 0x7fffffff, NONE_A = 1 << 20;
  const int lane = threadIdx.x;
  const bool cand = lane < T;                 // lanes T.. only help with the copies
  const int grp = cand ? lane / U : 0;        // my slot
  const int c = lane % U;                     // my candidate column within the slot
  const int first = grp * U;                  // the slot's first lane
  const bool lead = cand && c == 0;
  lds_init_wtab(b.wk_ovf + (size_t)blockIdx.x * (T * OVF_WORDS));
  MSB_AS_LDS u32x4* priv = (MSB_AS_LDS u32x4*)(uintptr_t)L::PRIV;
  MSB_AS_LDS u32x4* par_all = (MSB_AS_LDS u32x4*)(uintptr_t)L::PAR;   // [G][SG]
  MSB_AS_LDS double* wf = (MSB_AS_LDS double*)(uintptr_t)(L::WF + grp * 240);
  MSB_AS_LDS double* cf = (MSB_AS_LDS double*)(uintptr_t)(L::CF + (cand ? lane : 0) * 80);
  ParEngine pe;
  CandEngine ce;

  // the games of this wavefront's slots: kernels.h "Persistent wavefronts", a slot where k_play has a wavefront
  int* mine = b.pop + parity * POP_PARTS * POP_STRIDE;
  int* other = b.pop + (parity ^ 1) * POP_PARTS * POP_STRIDE;
  if (persistent && blockIdx.x == 0 && lane < POP_PARTS) other[lane * POP_STRIDE] = 0;
  const int part = blockIdx.x % POP_PARTS, rank = blockIdx.x / POP_PARTS;
  const int waves = ((int)gridDim.x - part + POP_PARTS - 1) / POP_PARTS;
  const int lo = persistent ? (int)((long long)n * part / POP_PARTS) : 0;
  const int hi = persistent ? (int)((long long)n * (part + 1) / POP_PARTS) : n;
  const int slots = waves * G;
  int t = NONE_G;   // the next game of my slot
  if (cand) {
    t = persistent ? lo + rank * G + grp : (int)blockIdx.x * G + grp;
    if (t >= hi) t = NONE_G;
  }
  int g = -1;       // the game my slot is playing
  GameMeta meta;
  __builtin_memset(&meta, 0, sizeof(meta));
  int left = 0;     // decisions my slot may still play of it in this launch
  bool have_before = false;
  double last_score = NAN;
  int played = 0;

  for (;;) {
    // ---- slots without a game take the next one; games that are over or out of rounds go back to HBM -------------------
    for (;;) {
      const bool take = cand && g < 0 && t != NONE_G;
      const unsigned long long takers = __ballot(take && lead);
      int nx = NONE_G;
      if (take && persistent && lead) nx = lo + slots + atomicAdd(&mine[part * POP_STRIDE], 1);
      nx = __shfl(nx, first);
      if (take) {
        g = t;
        t = nx >= hi ? NONE_G : nx;
        meta = b.meta[g];
        left = rounds;
        have_before = false;
        last_score = NAN;
        played = 0;
      }
      for (int i = 0; i < G; i++)
        if ((takers >> (i * U)) & 1) {   // one coalesced 16-B-per-lane pass per staged record
          const int gi = __builtin_amdgcn_readlane(g, i * U);
          const u32x4* src = (const u32x4*)(b.state + (size_t)gi * SW);
          for (int k = lane; k < SG; k += 64) par_all[i * SG + k] = src[k];
        }
      __syncthreads();
      if (take && lead) attach_rng(pe, b, g, meta.rng);
      __syncthreads();
      bool fin = false;
      if (cand && g >= 0) {
        if (take && meta.result != -2) {   // not a running game: nothing to play, nothing to write back
          if (lead) {
            b.meta[g].last_action = 255;
            if (b.best) b.best[g] = NAN;
          }
          g = -1;
        } else if (left == 0 || meta.result != -2) {   // out of rounds, or ended by a fault in its last decision
          fin = true;
        } else if (pe.have_winner() || meta.steps >= max_turns) {   // rollout contract
          const int b0 = pe.pl_base(0), b1 = pe.pl_base(1);
          int res = -1;
          if (pe.have_winner()) res = (b1 < 0 && b0 >= 0) ? 0 : (b0 < 0 && b1 >= 0) ? 1 : -1;
          meta.result = (int8_t)res;
          if (pe.have_winner()) meta.flags |= 1;
          if (played == 0) {
            meta.last_action = 255;
            last_score = NAN;
          }
          fin = true;
        }
      }
      const unsigned long long fins = __ballot(fin && lead);
      for (int i = 0; i < G; i++)
        if ((fins >> (i * U)) & 1) {
          const int gi = __builtin_amdgcn_readlane(g, i * U);
          u32x4* dst = (u32x4*)(b.state + (size_t)gi * SW);
          for (int k = lane; k < SG; k += 64) dst[k] = par_all[i * SG + k];
        }
      if (fin) {
        if (lead) {
          b.meta[g] = meta;
          if (b.best) b.best[g] = last_score;
        }
        g = -1;
      }
      __syncthreads();   // the LDS image is reused by the next game
      if (!__ballot(cand && g < 0 && t != NONE_G)) break;
    }
    if (!__ballot(cand && g >= 0)) break;   // every slot is out of games
    const bool on = cand && g >= 0;

    // ---- one decision of every slot that has a game (kernels.h play_game, per slot) -----------------------------------------
    uint64_t mask[3] = {0, 0, 0};
    if (on) {
      const msb_u64x4 lm = pe.legal_mask_v();
      mask[0] = lm[0];
      mask[1] = lm[1];
      mask[2] = lm[2];
    }
    uint64_t rem[3] = {mask[0], mask[1], mask[2]};
    const int n_legal = __popcll(mask[0]) + __popcll(mask[1]) + __popcll(mask[2]);
    const bool before_raises = on && pe.observation_raises();
    if (on) {
      const double* wt = b.weights + (size_t)(pe.local() == 0 ? meta.p1 : meta.p2) * 10;
      for (int i = c; i < 10; i += U) wf[i] = wt[i];
      if (!before_raises && !have_before && c == 0) pe.features(wf + 10);
    }
    __syncthreads();

    double run_s = 0.0;
    int run_a = NONE_A;
    int cfault = 0, feat_ok = 0, la_fault = 0, wl = 0;
    const bool multi = n_legal > U;
    for (int base = 0; __ballot(on && base < n_legal) != 0; base += U) {
      const int k = base + c;
      int n_act = 0;   // my slot's candidates in this pass
      if (on && base < n_legal) n_act = n_legal - base < U ? n_legal - base : U;
      // copy.deepcopy for every candidate of every slot, by all 64 lanes: granule idx of the interleaved image belongs to
      // column idx % T, which is the same column every time round for a lane (T divides 64)
      {
        const int col = lane % T;
        const int na = __shfl(n_act, (col / U) * U);
        __syncthreads();
        if ((col % U) < na)
          for (int idx = lane; idx < SG * T; idx += 64) priv[idx] = par_all[(col / U) * SG + idx / T];
        __syncthreads();
      }
      const bool active = on && k < n_legal;
      int a = NONE_A;
      if (active) a = nth_set_bit(rem, c);
      {   // this pass's actions leave the remaining set: everything up to the slot's last candidate
        const int a_last = __shfl(a, first + (n_act > 0 ? n_act - 1 : 0));
        if (n_act > 0)
          for (int w = 0; w < 3; w++) {
            const int rel = a_last - 64 * w;
            if (rel >= 63) rem[w] = 0;
            else if (rel >= 0) rem[w] &= ~((2ull << rel) - 1ull);
          }
      }
      double s = 0.0;   // except Exception -> 0.0 (evo/heuristic_agent.py:48-51)
      int my_fault = 0, my_feat = 0, f = 0;
      bool raises = false;
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
      {   // a look-ahead that hit a limit of this build: the slot's first such lane marks the game (meta.la_fault)
        const unsigned long long lfb = __ballot(active && f >= FAULT_CAPACITY);
        const unsigned bits = (unsigned)(lfb >> first) & ((1u << U) - 1u);
        const int fv = __shfl(f, first + (bits ? __builtin_ctz(bits) : 0));
        if (on && bits && !la_fault) la_fault = fv;
      }
      // first maximum over the ascending legal list == (max score, then min action id): a butterfly over the slot's U
      // lanes leaves the slot's result in every one of them
      double cs = s;
      int ca = a;
      for (int off = U / 2; off >= 1; off >>= 1) {
        const double os = __shfl_xor(cs, off);
        const int oa = __shfl_xor(ca, off);
        const bool better = (oa != NONE_A) && (ca == NONE_A || os > cs || (os == cs && oa < ca));
        if (better) {
          cs = os;
          ca = oa;
        }
      }
      const unsigned long long winners = __ballot(active && a == ca);
      const unsigned wbits = (unsigned)(winners >> first) & ((1u << U) - 1u);
      const int wl_new = first + (wbits ? __builtin_ctz(wbits) : 0);
      const int cfault_new = __shfl(my_fault, wl_new), feat_new = __shfl(my_feat, wl_new);
      const bool nb = on && ca != NONE_A && (run_a == NONE_A || cs > run_s);   // later passes hold larger action ids: strict >
      if (nb) {
        run_s = cs;
        run_a = ca;
        wl = wl_new;
        cfault = cfault_new;
        feat_ok = feat_new;
        if (feat_ok)   // the winner keeps its features: the next decision's "before" side
          for (int i = c; i < 10; i += U) wf[20 + i] = ((MSB_AS_LDS const double*)(uintptr_t)L::CF)[wl * 10 + i];
      }
      const unsigned long long parks = __ballot(nb && multi && lead);
      if (parks) {   // the best successor so far waits in the game's record in HBM while later passes reuse its column
        __syncthreads();
        for (int i = 0; i < G; i++)
          if ((parks >> (i * U)) & 1) {
            const int gi = __builtin_amdgcn_readlane(g, i * U), wi = __builtin_amdgcn_readlane(wl, i * U);
            u32x4* dst = (u32x4*)(b.state + (size_t)gi * SW);
            for (int kk = lane; kk < SG; kk += 64) dst[kk] = priv[kk * T + wi];
          }
      }
    }
    __syncthreads();
    // commit: adapter = adapter.apply_action(best).  The successor carries its own stream cursor (H_RNGPOS).
    {
      const unsigned long long ons = __ballot(on && lead);
      for (int i = 0; i < G; i++)
        if ((ons >> (i * U)) & 1) {
          const int gi = __builtin_amdgcn_readlane(g, i * U), wi = __builtin_amdgcn_readlane(wl, i * U);
          if (__builtin_amdgcn_readlane((int)multi, i * U)) {
            // what this wavefront parked: agent-scope loads, so that the per-CU vector cache cannot answer with an older line
            unsigned long long* src = (unsigned long long*)(b.state + (size_t)gi * SW);
            for (int kk = lane; kk < 2 * SG; kk += 64)
              ((MSB_AS_LDS unsigned long long*)par_all)[i * 2 * SG + kk] = __hip_atomic_load(src + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            for (int kk = lane; kk < SG; kk += 64) par_all[i * SG + kk] = priv[kk * T + wi];
          }
        }
    }
    __syncthreads();
    {
      uint32_t new_pos = on ? (uint32_t)pe.rng_pos() : 0u;
      int cur = (meta.rng >> 16) & 1;
      const bool refill = on && new_pos >= (uint32_t)MT_N;
      const unsigned long long refs = __ballot(refill && lead);
      for (int i = 0; i < G; i++)
        if ((refs >> (i * U)) & 1)   // the used-up block becomes the new "next" block
          wave_refill(b, __builtin_amdgcn_readlane(g, i * U), __builtin_amdgcn_readlane(cur, i * U), (MSB_AS_LDS uint32_t*)priv, lane);
      if (refill) {
        new_pos -= MT_N;
        cur ^= 1;
        if (lead) pe.rng_block_advance();
      }
      if (on) {
        meta.rng = new_pos | ((uint32_t)cur << 16);
        if (lead) attach_rng(pe, b, g, meta.rng);
        // the committed successor's features become the "before" side of the next decision
        have_before = feat_ok != 0;
        if (have_before)
          for (int i = c; i < 10; i += U) wf[10 + i] = wf[20 + i];
      }
      __syncthreads();
    }
    if (on) {
      meta.steps++;
      meta.last_action = (uint8_t)run_a;
      meta.lookahead += (uint32_t)n_legal;
      meta.decided++;
      if (la_fault && !meta.la_fault) meta.la_fault = (uint8_t)la_fault;
      last_score = run_s;
      played++;
      left--;
      if (cfault) {   // evo/fitness.py:208-210: an exception while applying the action ends the game as a draw
        meta.fault = (uint8_t)cfault;
        meta.result = -1;
      }
    }
  }
}

}  // namespace msbk
