// Stormbound rules core: one game step, restated from the reference's object engine for a
// flat state record (state.h).  Compiled for gfx950 (product) and for the host (oracle).
//
// Every function cites the reference code it restates.  The reference's bugs are reproduced on
// purpose: targeted spells land one tile late, Player.opponent is
// `self` on every second turn, statuses are multisets, dead units keep walking a cached path,
// turn-start movement iterates a snapshot of entity objects, and deferred triggers are a LIFO
// stack with a non-counting re-entrancy latch.
//
// Control flow: the reference recurses (move -> ability -> deal_damage -> destroy -> ability -> command -> move ...).
// Here every function on such a cycle is a FRAME on an explicit per-game work stack and Engine::run() is the only
// loop (see "Control flow" below): no call cycle is left in the C++, so the device code needs no dynamic stack.
#pragma once
#include "mt19937.h"
#include "pyset.h"
#include "state.h"

#if defined(MSB_FAULT_TRACE) && !defined(__HIPCC__)
#include <execinfo.h>
#include <stdio.h>
static inline void msb_fault_trace(int code) {
  void* bt[24];
  int n = backtrace(bt, 24);
  fprintf(stderr, "set_fault(%d)\n", code);
  backtrace_symbols_fd(bt, n, 2);
}
#endif

#if defined(MSB_COUNT_FRAMES) && !defined(__HIPCC__)
#define MSB_COUNT_EVENT(i_) msb_frame_count[i_]++
static long long msb_frame_count[64];   // study build of the host library: handler invocations per frame type, [0] = all, [15] = deepest stack, [16..27] = histogram of sp / 4, [32..47] = F_MOVE by state, [48..49] = F_RUNAB by state, [50..] = events inside move()
#else
#define MSB_COUNT_EVENT(i_)
#endif

namespace msb {

struct P {
  int x, y;
};
MSB_HD MSB_INL bool p_valid(P p) { return p.x >= 0 && p.x <= 3 && p.y >= 0 && p.y <= 4; }        // point.py:15-17
MSB_HD MSB_INL bool p_is_base(P p) { return p.x == -1 && (p.y == -1 || p.y == 5); }              // point.py:19-21
MSB_HD MSB_INL bool p_eq(P a, P b) { return a.x == b.x && a.y == b.y; }
// x in [-1,6], y in [-1,5]; points further out (a path running on past a base) only ever get compared, never decoded --
// unsigned arithmetic keeps the shift of their negative row defined
MSB_HD MSB_INL int p_pack(P p) { return (int)(((unsigned)(p.y + 1) << 3) | (unsigned)(p.x + 1)); }
MSB_HD MSB_INL P p_unpack(int v) { return P{(v & 7) - 1, (v >> 3) - 1}; }
MSB_HD MSB_INL int p_tile(P p) { return p.y * 4 + p.x; }
MSB_HD MSB_INL P tile_p(int t) { return P{t & 3, t >> 2}; }
constexpr int PK_NONE = 0xFF;

// Point lists (results of the selectors: at most 20 tiles + 2 bases) live in REGISTERS: 24 six-bit
// packed points in three 64-bit lanes of a 4 x u64 vector, the length in the fourth.  A vector type is
// passed to and returned from the non-inlined selector functions in VGPRs; a byte array would sit in
// the per-lane scratch stack and every access would be a global-memory round trip (the round-1 profile
// showed 72 % of the wave cycles waiting on exactly that).
typedef unsigned long long msb_u64x4 __attribute__((vector_size(32)));   // <4 x i64>: GCC and clang spelling
struct PList {
  msb_u64x4 w;
  MSB_HD MSB_INL int n() const { return (int)w[3]; }
  MSB_HD MSB_INL void set_n(int k) { w[3] = (unsigned long long)k; }
  MSB_HD MSB_INL void clear() { w = msb_u64x4{0ull, 0ull, 0ull, 0ull}; }
  MSB_HD MSB_INL int get(int i) const {   // packed point at index i
    unsigned long long x = i < 10 ? w[0] : (i < 20 ? w[1] : w[2]);
    int k = i < 10 ? i : (i < 20 ? i - 10 : i - 20);
    return (int)((x >> (6 * k)) & 63ull);
  }
  MSB_HD MSB_INL void set(int i, int v) {
    int k = i < 10 ? i : (i < 20 ? i - 10 : i - 20);
    unsigned long long msk = ~(63ull << (6 * k)), bits = (unsigned long long)(v & 63) << (6 * k);
    if (i < 10)
      w[0] = (w[0] & msk) | bits;
    else if (i < 20)
      w[1] = (w[1] & msk) | bits;
    else
      w[2] = (w[2] & msk) | bits;
  }
  // the same three words as 24 one-byte entries (entity slot ids)
  MSB_HD MSB_INL int get8(int i) const {
    unsigned long long x = i < 8 ? w[0] : (i < 16 ? w[1] : w[2]);
    return (int)((x >> (8 * (i & 7))) & 0xffull);
  }
  MSB_HD MSB_INL void set8(int i, int v) {
    unsigned long long msk = ~(0xffull << (8 * (i & 7))), bits = (unsigned long long)(v & 0xff) << (8 * (i & 7));
    if (i < 8)
      w[0] = (w[0] & msk) | bits;
    else if (i < 16)
      w[1] = (w[1] & msk) | bits;
    else
      w[2] = (w[2] & msk) | bits;
  }
  MSB_HD MSB_INL void push_raw(int v) {
    int i = n();
    if (i < 24) {
      set(i, v);
      set_n(i + 1);
    }
  }
  MSB_HD MSB_INL void push(P p) { push_raw(p_pack(p)); }
  MSB_HD MSB_INL P at(int i) const { return p_unpack(get(i)); }
  MSB_HD MSB_INL bool has(P p) const {
    int k = p_pack(p), m = n();
    for (int i = 0; i < m; i++)
      if (get(i) == k) return true;
    return false;
  }
  MSB_HD MSB_INL void remove_at(int i) {
    int m = n();
    for (int j = i; j + 1 < m; j++) set(j, get(j + 1));
    set_n(m - 1);
  }
};

// Target descriptor (target.py:18-29) packed into one 64-bit scalar (passed in registers):
//   kind 0-1 | side 2-3 | non_hero 4 | base 5 | status 6-10 | xstatus 11-15 | types 16-31 | xtypes 32-47 |
//   limit+32768 48-63 (0 = None)
typedef unsigned long long Tgt;
constexpr int LIMIT_NONE = -32768;
MSB_HD MSB_INL Tgt mk_tgt(int kind, int side) { return (Tgt)kind | ((Tgt)side << 2); }
MSB_HD MSB_INL Tgt tgt_types(Tgt t, int types) { return t | ((Tgt)(types & 0xffff) << 16); }
MSB_HD MSB_INL Tgt tgt_xtypes(Tgt t, int xtypes) { return t | ((Tgt)(xtypes & 0xffff) << 32); }
MSB_HD MSB_INL Tgt tgt_status(Tgt t, int status) { return t | ((Tgt)(status & 31) << 6); }
MSB_HD MSB_INL Tgt tgt_xstatus(Tgt t, int xstatus) { return t | ((Tgt)(xstatus & 31) << 11); }
MSB_HD MSB_INL Tgt tgt_base(Tgt t) { return t | (1ull << 5); }
MSB_HD MSB_INL Tgt tgt_non_hero(Tgt t) { return t | (1ull << 4); }
MSB_HD MSB_INL Tgt tgt_limit(Tgt t, int limit) { return (t & 0x0000ffffffffffffull) | ((Tgt)((limit + 32768) & 0xffff) << 48); }
MSB_HD MSB_INL int tg_kind(Tgt t) { return (int)(t & 3); }
MSB_HD MSB_INL int tg_side(Tgt t) { return (int)((t >> 2) & 3); }
MSB_HD MSB_INL bool tg_non_hero(Tgt t) { return (t >> 4) & 1; }
MSB_HD MSB_INL bool tg_base(Tgt t) { return (t >> 5) & 1; }
MSB_HD MSB_INL int tg_status(Tgt t) { return (int)((t >> 6) & 31); }
MSB_HD MSB_INL int tg_xstatus(Tgt t) { return (int)((t >> 11) & 31); }
MSB_HD MSB_INL int tg_types(Tgt t) { return (int)((t >> 16) & 0xffff); }
MSB_HD MSB_INL int tg_xtypes(Tgt t) { return (int)((t >> 32) & 0xffff); }
MSB_HD MSB_INL int tg_limit(Tgt t) { return (int)((t >> 48) & 0xffff) - 32768; }   // LIMIT_NONE when unset
MSB_HD MSB_INL Tgt mk_tgt(const TargetSpec& s) {
  Tgt t = mk_tgt(s.kind, s.side);
  t = tgt_types(t, s.types);
  t = tgt_xtypes(t, s.xtypes);
  if (s.limit >= 0) t = tgt_limit(t, s.limit);
  if (s.non_hero) t = tgt_non_hero(t);
  t = tgt_status(t, s.status);
  t = tgt_xstatus(t, s.xstatus);
  if (s.base) t = tgt_base(t);
  return t;
}

// Results of Board.at (board.py:58-65)
constexpr int AT_NONE = -1;
constexpr int AT_PLAYER = 256;  // + PlayerOrder (above every entity slot id)

enum : int { SH_FRONT, SH_BEHIND, SH_SIDE, SH_ROW, SH_COLUMN, SH_BORDERING, SH_SURROUNDING };

#if defined(MSB_CAP_DEPTH)
constexpr int MAX_DEPTH = MSB_CAP_DEPTH;   // capacity studies
#else
constexpr int MAX_DEPTH = 40;
#endif

// Which rules-core functions are inlined into their callers was settled by same-box A/B runs (scripts/ab_env.sh,
// scripts/ab_libs.sh; 65 536 games).  Rounds 1-2, on the recursive core: new_entity + set_path + calculate_front_line inlined
// +2.4 %; get_targets +7.8 %, shape_targets +5 %, draw +1.2 %; shape_tiles, shuffle/sorted_head, legal_mask_v and the
// command/teleport/push_pull/spawn helpers gained nothing and stay out of line; one non-inlined function per card instead of
// one switch function +3 %.  Round 3, on the work-stack core: the handlers of the rare frames and the abilities that call
// back into the core are out of line (they would otherwise sit inside run()'s loop and raise every lane's register count);
// step_impl out of line 1 315 M env-steps/s against 1 277 inlined (DESIGN.md section 4 has the trade: 800 B of saved
// registers per pass against 96 VGPRs).
#ifndef MSB_A_NEWENT
#define MSB_A_NEWENT MSB_INL
#endif
#ifndef MSB_A_SETPATH
#define MSB_A_SETPATH MSB_INL
#endif
#ifndef MSB_A_FRONT
#define MSB_A_FRONT MSB_INL
#endif
#ifndef MSB_A_TARGETS
#define MSB_A_TARGETS MSB_INL
#endif
#ifndef MSB_A_DRAW
#define MSB_A_DRAW MSB_INL
#endif
#ifndef MSB_A_SHAPE
#define MSB_A_SHAPE MSB_INL
#endif
// handlers of the rare frames, abilities with nested calls: out of line
#ifndef MSB_A_RARE
#define MSB_A_RARE MSB_NOINLINE
#endif
// leaf card abilities: one non-inlined function each (a call from run(), no call inside)
#ifndef MSB_A_CARD
#define MSB_A_CARD MSB_NOINLINE
#endif
#ifndef MSB_A_TILES
#define MSB_A_TILES MSB_NOINLINE
#endif
#ifndef MSB_A_SHUFFLE
#define MSB_A_SHUFFLE MSB_NOINLINE
#endif
#ifndef MSB_A_LEGAL
#define MSB_A_LEGAL MSB_NOINLINE
#endif
#ifndef MSB_A_MISC
#define MSB_A_MISC MSB_NOINLINE
#endif
#ifndef MSB_A_STEP
#define MSB_A_STEP MSB_NOINLINE
#endif
#ifndef MSB_A_TURN
#define MSB_A_TURN MSB_NOINLINE
#endif

template <class M>
struct Engine {
  M m;

  // ---- numpy legacy RandomState draws over the record's stream window (mt19937.h) ----------------
  MSB_HD MSB_INL uint32_t rng_next_u32() {
    if (REM_LISTS && ctx() != 0) return world_rng_next();   // an entity of a frozen world draws from THAT world's stream copy
    uint32_t i = (uint32_t)m.ld16(H_RNGPOS) & 0xffffu;
    if (i < (uint32_t)(2 * MT_N)) {
      uint64_t blk = m.ld64(i < (uint32_t)MT_N ? H_RNGCUR : H_RNGNXT);
      if (blk == 0) {   // no stream window attached to this record: flag it, never dereference
        m.st8(H_RNGOVER, 1);
        return 0;
      }
      m.st16(H_RNGPOS, (int)(i + 1));
      return ((MSB_RNG_PTR)(uintptr_t)blk)[i < (uint32_t)MT_N ? i : i - MT_N];
    }
    m.st8(H_RNGOVER, 1);   // a single step would have to draw more than 624 words
    return 0;
  }
  // rk_interval / buffered_bounded_masked_uint32 for max <= 0xffffffff
  MSB_HD MSB_INL uint32_t rng_interval(uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
      v = rng_next_u32() & mask;
    } while (v > max && !m.ld8(H_RNGOVER));
    return v;
  }
  MSB_HD MSB_INL int rng_randint(int lo, int hi) { return lo + (int)rng_interval((uint32_t)(hi - 1 - lo)); }
  MSB_HD MSB_INL double rng_random_sample() {
    uint32_t a = rng_next_u32() >> 5, b = rng_next_u32() >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
  }
  MSB_HD MSB_INL void rng_attach(const uint32_t* cur, const uint32_t* nxt, uint32_t pos) {
    m.st64(H_RNGCUR, (uint64_t)(uintptr_t)cur);
    m.st64(H_RNGNXT, (uint64_t)(uintptr_t)nxt);
    m.st16(H_RNGPOS, (int)pos);
    m.st8(H_RNGOVER, 0);
  }
  MSB_HD MSB_INL uint32_t rng_pos() const { return (uint32_t)m.ld16(H_RNGPOS) & 0xffffu; }

  // ------------------------------------------------------------------------------------------
  // raw field access
  // ------------------------------------------------------------------------------------------
  MSB_HD MSB_INL int fault() const { return m.ld8(H_FAULT); }
  MSB_HD MSB_INL void set_fault(int code) {
#if defined(MSB_FAULT_TRACE) && !defined(__HIPCC__)
    msb_fault_trace(code);   // debugging aid of the host build (oracle/Makefile: make trace)
#endif
    if (m.ld8(H_FAULT) == 0) m.st8(H_FAULT, code);
  }
  MSB_HD MSB_INL int local() const { return m.ld8(H_TOPLAY); }     // order of board.local
  MSB_HD MSB_INL int remote() const { return m.ld8(H_TOPLAY) ^ 1; }
  MSB_HD MSB_INL int cp() const { return m.ld8(H_CP); }            // order of board.current_player
  MSB_HD MSB_INL int phase() const { return m.ld8(H_PHASE); }

  MSB_HD MSB_INL int pl(int order, int f) const { return OFF_PL + order * PL_SIZE + f; }
  MSB_HD MSB_INL int pl_base(int o) const { return m.ld16g(pg(o) + (P_BASE >> 4), P_BASE & 15); }
  MSB_HD MSB_INL void set_pl_base(int o, int v) { m.st16g(pg(o) + (P_BASE >> 4), P_BASE & 15, v); }
  MSB_HD MSB_INL int pl_mana(int o) const { return m.ld16g(pg(o) + (P_MANA >> 4), P_MANA & 15); }
  MSB_HD MSB_INL void set_pl_mana(int o, int v) { m.st16g(pg(o) + (P_MANA >> 4), P_MANA & 15, v); }
  MSB_HD MSB_INL int pl_maxmana(int o) const { return m.ld16g(pg(o) + (P_MAXMANA >> 4), P_MAXMANA & 15); }
  MSB_HD MSB_INL int pl_front(int o) const { return m.ld8g(pg(o) + (P_FRONT >> 4), P_FRONT & 15); }
  MSB_HD MSB_INL void set_pl_front(int o, int v) { m.st8g(pg(o) + (P_FRONT >> 4), P_FRONT & 15, v); }
  MSB_HD MSB_INL int pl_hand_n(int o) const { return m.ld8g(pg(o) + (P_HAND_N >> 4), P_HAND_N & 15); }
  MSB_HD MSB_INL int pl_deck_n(int o) const { return m.ld8g(pg(o) + (P_DECK_N >> 4), P_DECK_N & 15); }
  // ---- hand / deck lists ---------------------------------------------------------------------
  // hand_ref/deck_ref: byte offset of the card-instance record {card, cost, flags, x} at a list position.
  // Standard record: values stored in place.  Extended record: positions hold object ids into the player's
  // instance table (state.h).
#if defined(MSB_EXT) && MSB_EXT
  MSB_HD MSB_INL int hand_id(int o, int i) const { return m.ld8(pl(o, P_HAND + i)); }
  MSB_HD MSB_INL int deck_id(int o, int i) const { return m.ld8(pl(o, P_DECK + i)); }
  MSB_HD MSB_INL int hand_ref(int o, int i) const { return pl(o, P_INST + 4 * hand_id(o, i)); }
  MSB_HD MSB_INL int deck_ref(int o, int i) const { return pl(o, P_INST + 4 * deck_id(o, i)); }
  // a free object id: referenced by neither list
  MSB_HD MSB_NOINLINE int inst_alloc(int o) {
    static_assert(INST_CAP <= 64, "one 64-bit word of object ids");
    uint64_t used = 0;
    for (int i = 0; i < pl_hand_n(o); i++) used |= 1ull << hand_id(o, i);
    for (int i = 0; i < pl_deck_n(o); i++) used |= 1ull << deck_id(o, i);
    for (int k = 0; k < INST_CAP; k++)
      if (!(used & (1ull << k))) return k;
    set_fault(FAULT_CAP_DECK);
    return 0;
  }
#else
  MSB_HD MSB_INL int hand_ref(int o, int i) const { return pl(o, P_HAND + 4 * i); }
  MSB_HD MSB_INL int deck_ref(int o, int i) const { return pl(o, P_DECK + 4 * i); }
#endif
  MSB_HD MSB_INL int hand_card(int o, int i) const { return m.ld8(hand_ref(o, i)); }
  MSB_HD MSB_INL int hand_cost(int o, int i) const { return m.ld8(hand_ref(o, i) + 1); }
  MSB_HD MSB_INL int hand_flags(int o, int i) const { return m.ld8(hand_ref(o, i) + 2); }
  MSB_HD MSB_INL int hand_x(int o, int i) const { return m.ld8(hand_ref(o, i) + 3); }
  MSB_HD MSB_INL int deck_card(int o, int i) const { return m.ld8(deck_ref(o, i)); }
  MSB_HD MSB_INL int deck_cost(int o, int i) const { return m.ld8(deck_ref(o, i) + 1); }
  MSB_HD MSB_INL int deck_flags(int o, int i) const { return m.ld8(deck_ref(o, i) + 2); }
  MSB_HD MSB_INL int deck_x(int o, int i) const { return m.ld8(deck_ref(o, i) + 3); }
  // weight slot of deck position i (standard) / of the object listed there (extended), in (granule, offset) form
#if defined(MSB_EXT) && MSB_EXT
  MSB_HD MSB_INL int deck_wslot(int o, int i) const { return deck_id(o, i); }
#else
  MSB_HD MSB_INL int deck_wslot(int o, int i) const { return i; }
#endif
  static_assert(E_PATH % 16 == 0, "granule-aligned arrays");
  // Card.weight of the card at deck position i = wtab[age] (state.h: the record keeps the age, not the f64)
  MSB_HD MSB_INL int deck_age(int o, int i) const { return m.ld8(pl(o, P_AGE + deck_wslot(o, i))); }
  MSB_HD MSB_INL void set_deck_age(int o, int i, int a) { m.st8(pl(o, P_AGE + deck_wslot(o, i)), a); }
  MSB_HD MSB_INL double deck_w(int o, int i) const { return M::wtab(deck_age(o, i)); }
  // Player.reweight, player.py:57-59: card.weight = card.weight * 1.6 + 100 for every deck entry = age + 1
  MSB_HD MSB_INL void reweight(int o) {
    int n = pl_deck_n(o);
#if defined(MSB_EXT) && MSB_EXT
    for (int i = 0; i < n; i++) {   // per list position: an object listed twice ages twice
      int a = deck_age(o, i);
      if (a >= AGE_MAX) {
        set_fault(FAULT_CAP_DECK);
        return;
      }
      set_deck_age(o, i, a + 1);
    }
#else
    static_assert(DECK_CAP == 12, "three age words");
    for (int w = 0; w < 3; w++) {   // four ages per 32-bit word
      int k = n - 4 * w;
      if (k <= 0) break;
      uint32_t one = k >= 4 ? 0x01010101u : (0x01010101u >> (8 * (4 - k)));
      uint32_t v = m.ld32(pl(o, P_AGE + 4 * w));
      uint32_t t = ~v;   // a byte of v is 0xFF <=> that byte of t is 0
      if (((t - 0x01010101u) & ~t & 0x80808080u) != 0) {
        set_fault(FAULT_CAP_DECK);
        return;
      }
      m.st32(pl(o, P_AGE + 4 * w), v + one);
    }
#endif
  }
  // packed path of entity e (u32[NUM_ENT] after the entity granules)
  MSB_HD MSB_INL uint32_t e_path(int e) const { return m.ld32g(E_PATH / 16 + (e >> 2), 4 * (e & 3)); }
  MSB_HD MSB_INL void e_set_path(int e, uint32_t v) { m.st32g(E_PATH / 16 + (e >> 2), 4 * (e & 3), v); }
  // strength attribute of a card instance in hand/deck (see CF_ALIAS / CF_STR in state.h)
  MSB_HD MSB_INL int inst_strength(int card, int fl, int x) const {
    if (fl & CF_ALIAS) return m.ld16g(eg(x), EO_STR);
    if (fl & (CF_STR | CF_XBASE)) return x;
    return card < NUM_CARDS ? g_cards[card].strength : 0;
  }
  // list primitives -----------------------------------------------------------------------------
  MSB_HD MSB_INL void deck_remove_at(int o, int j) {   // del deck[j]
    int n = pl_deck_n(o);
#if defined(MSB_EXT) && MSB_EXT
    for (int i = j; i + 1 < n; i++) m.st8(pl(o, P_DECK + i), m.ld8(pl(o, P_DECK + i + 1)));
#else
    for (int i = j; i + 1 < n; i++) {
      m.st32(pl(o, P_DECK + 4 * i), m.ld32(pl(o, P_DECK + 4 * (i + 1))));
      m.st8(pl(o, P_AGE + i), m.ld8(pl(o, P_AGE + i + 1)));
    }
    m.st8(pl(o, P_AGE + n - 1), 0);   // ages beyond the list stay 0 (reweight adds to whole words)
#endif
    m.st8(pl(o, P_DECK_N), n - 1);
  }
  MSB_HD MSB_INL void hand_remove_at(int o, int j) {   // del hand[j]
    int n = pl_hand_n(o);
#if defined(MSB_EXT) && MSB_EXT
    for (int i = j; i + 1 < n; i++) m.st8(pl(o, P_HAND + i), m.ld8(pl(o, P_HAND + i + 1)));
#else
    for (int i = j; i + 1 < n; i++) m.st32(pl(o, P_HAND + 4 * i), m.ld32(pl(o, P_HAND + 4 * (i + 1))));
#endif
    m.st8(pl(o, P_HAND_N), n - 1);
  }
  // hand.append(deck[idx]) (the object keeps living in the deck list until deck.remove)
  MSB_HD MSB_INL void hand_push_from_deck(int o, int idx) {
    int hn = pl_hand_n(o);
    if (hn >= HAND_CAP) {
      set_fault(FAULT_CAP_HAND);
      return;
    }
#if defined(MSB_EXT) && MSB_EXT
    m.st8(pl(o, P_HAND + hn), deck_id(o, idx));
#else
    m.st32(pl(o, P_HAND + 4 * hn), m.ld32(pl(o, P_DECK + 4 * idx)));
#endif
    m.st8(pl(o, P_HAND_N), hn + 1);
  }
  // A handle names a card object that has just left a list: the value itself (standard) or its id (extended).
  MSB_HD MSB_INL uint32_t hand_handle(int o, int i) const {
#if defined(MSB_EXT) && MSB_EXT
    return (uint32_t)hand_id(o, i);
#else
    return m.ld32(pl(o, P_HAND + 4 * i));
#endif
  }
  // deck.append(obj).  Standard record: a card coming from the hand always has weight 1 (player.py:50).
  // Extended record: the object keeps whatever weight it has (it may also still be listed in the deck).
  MSB_HD MSB_INL void deck_push_handle(int o, uint32_t h) {
    int n = pl_deck_n(o);
    if (n >= DECK_CAP) {
      set_fault(FAULT_CAP_DECK);
      return;
    }
#if defined(MSB_EXT) && MSB_EXT
    m.st8(pl(o, P_DECK + n), (int)h);
#else
    m.st32(pl(o, P_DECK + 4 * n), h);
    m.st8(pl(o, P_AGE + n), 0);
#endif
    m.st8(pl(o, P_DECK_N), n + 1);
  }
  // a brand-new card object (ua20's copy, b305's returning structure) appended to the deck / hand
  MSB_HD MSB_NOINLINE void push_new_instance(int o, bool to_hand, int card, int cost, int fl, int x) {
    int n = to_hand ? pl_hand_n(o) : pl_deck_n(o);
    if (n >= (to_hand ? HAND_CAP : DECK_CAP)) {
      set_fault(to_hand ? FAULT_CAP_HAND : FAULT_CAP_DECK);
      return;
    }
    int rec, wrec = -1;
#if defined(MSB_EXT) && MSB_EXT
    int id = inst_alloc(o);
    if (fault()) return;
    m.st8(pl(o, (to_hand ? P_HAND : P_DECK) + n), id);
    rec = pl(o, P_INST + 4 * id);
    wrec = pl(o, P_AGE + id);
    m.st8(pl(o, P_IPOS + id), IPOS_UNKNOWN);
#else
    rec = pl(o, (to_hand ? P_HAND : P_DECK) + 4 * n);
    if (!to_hand) wrec = pl(o, P_AGE + n);
#endif
    m.st8(rec, card);
    m.st8(rec + 1, cost);
    m.st8(rec + 2, fl);
    m.st8(rec + 3, x);
    if (wrec >= 0) m.st8(wrec, 0);   // weight = 1
    m.st8(pl(o, to_hand ? P_HAND_N : P_DECK_N), n + 1);
  }

  MSB_HD MSB_INL int board_at(int tile) const { return m.ld8(OFF_BOARD + tile); }
  MSB_HD MSB_INL void board_put(int tile, int slot) { m.st8(OFF_BOARD + tile, slot); }

  MSB_HD MSB_INL static int ent(int e) { return OFF_ENT + ENT_SIZE * e; }
  // granule index of entity e / of player o's block: field accesses go through (granule, offset-in-granule), which
  // costs one address instruction on the lane-interleaved LDS image instead of four for a byte offset
  static_assert(OFF_ENT % 16 == 0 && ENT_SIZE == 16 && OFF_PL % 16 == 0 && PL_SIZE % 16 == 0, "granule-aligned blocks");
  MSB_HD MSB_INL static int eg(int e) { return OFF_ENT / 16 + e; }
  MSB_HD MSB_INL static int pg(int order) { return OFF_PL / 16 + order * (PL_SIZE / 16); }
  MSB_HD MSB_INL int e_card(int e) const { return m.ld8g(eg(e), EO_CARD); }
  MSB_HD MSB_INL int e_flags(int e) const { return m.ld8g(eg(e), EO_FLAGS); }
  MSB_HD MSB_INL int e_owner(int e) const { return m.ld8g(eg(e), EO_FLAGS) & EF_OWNER; }
  MSB_HD MSB_INL bool e_ff(int e) const { return (m.ld8g(eg(e), EO_FLAGS) & EF_FF) != 0; }
  MSB_HD MSB_INL void e_set_flag(int e, int bit, bool on) {
    int f = m.ld8g(eg(e), EO_FLAGS);
    m.st8g(eg(e), EO_FLAGS, on ? (f | bit) : (f & ~bit));
  }
  MSB_HD MSB_INL P e_pos(int e) const { return tile_p(m.ld8g(eg(e), EO_POS)); }
  MSB_HD MSB_INL void e_set_pos(int e, P p) { m.st8g(eg(e), EO_POS, p_tile(p)); }
  MSB_HD MSB_INL int e_mov(int e) const { return m.ld8g(eg(e), EO_MOV); }
  MSB_HD MSB_INL int e_str(int e) const { return m.ld16g(eg(e), EO_STR); }
  MSB_HD MSB_INL void e_set_str(int e, int v) { m.st16g(eg(e), EO_STR, v); }
  MSB_HD MSB_INL int e_dmg(int e) const { return m.ld16g(eg(e), EO_DMG); }
  MSB_HD MSB_INL void e_set_dmg(int e, int v) { m.st16g(eg(e), EO_DMG, v); }
  MSB_HD MSB_INL int e_st(int e, int s) const { return m.ld8g(eg(e), EO_ST + s); }
  MSB_HD MSB_INL void e_st_add(int e, int s) {
    int c = m.ld8g(eg(e), EO_ST + s);
    if (c >= 255) {
      set_fault(FAULT_STATUS_SAT);
      return;
    }
    m.st8g(eg(e), EO_ST + s, c + 1);
  }
  // list.remove(x) raises ValueError when x is absent
  MSB_HD MSB_INL void e_st_remove(int e, int s) {
    int c = m.ld8g(eg(e), EO_ST + s);
    if (c == 0) {
      set_fault(FAULT_PY_EXCEPTION);
      return;
    }
    m.st8g(eg(e), EO_ST + s, c - 1);
  }

  // ---- card statics (tokens: board.py:298-322) -------------------------------------------------
  MSB_HD MSB_INL bool card_is_unit(int c) const {
    return c < NUM_CARDS ? g_cards[c].kind == KIND_UNIT : (c >= TOKEN_UNIT_BASE && c < TOKEN_UNIT_BASE + 16);
  }
  MSB_HD MSB_INL bool card_is_struct(int c) const { return c < NUM_CARDS ? g_cards[c].kind == KIND_STRUCT : c == TOKEN_STRUCT; }
  MSB_HD MSB_INL int card_types(int c) const { return c < NUM_CARDS ? g_cards[c].types : (c < TOKEN_STRUCT ? (1 << (c - TOKEN_UNIT_BASE)) : 0); }
  MSB_HD MSB_INL int card_first_type(int c) const { return c < NUM_CARDS ? g_cards[c].first_type : c - TOKEN_UNIT_BASE; }
  MSB_HD MSB_INL int card_trigger(int c) const { return c < NUM_CARDS ? g_cards[c].trigger : TR_NONE; }
  MSB_HD MSB_INL bool card_has_ability(int c) const { return c < NUM_CARDS ? g_cards[c].has_ability != 0 : false; }
  // int(card), card.py:25-46.  Returns -1 where the reference raises ValueError.
  MSB_HD MSB_INL int card_int_id(int c) const {
    if (c < NUM_CARDS) return g_cards[c].int_id;
    if (c == TOKEN_STRUCT) return 1;                               // card_id "b001"
    int t = c - TOKEN_UNIT_BASE;                                   // "f" + str(t).zfill(3) parsed as hex
    return 0x4000 + (t < 10 ? t : 0x10 + (t - 10));
  }
  // static facts of the entity's card, cached in the entity record when it is created (no card-table
  // load -- a global-memory round trip -- inside the selectors and the movement loop)
  MSB_HD MSB_INL int e_kind(int e) const { return m.ld8g(eg(e), EO_KIND); }
  MSB_HD MSB_INL bool e_is_unit(int e) const { return (e_kind(e) & EK_UNIT) != 0; }
  MSB_HD MSB_INL bool e_has_ability(int e) const { return (e_kind(e) & EK_ABILITY) != 0; }
  MSB_HD MSB_INL int e_card_trigger(int e) const { return ((e_kind(e) >> 2) & 15) - 1; }   // card_trigger(e_card(e))
  MSB_HD MSB_INL int e_trigger(int e) const {  // Unit.trigger; Structures have no .trigger attribute
    int k = e_kind(e);
    return (k & EK_UNIT) ? ((k >> 2) & 15) - 1 : TR_NONE;
  }
  MSB_HD MSB_INL int kind_byte(int card) const {
    return (card_is_unit(card) ? EK_UNIT : 0) | (card_has_ability(card) ? EK_ABILITY : 0) | ((card_trigger(card) + 1) << 2);
  }
  MSB_HD MSB_INL bool e_disabled(int e) const { return e_st(e, ST_DISABLED) > 0; }
  MSB_HD MSB_INL bool e_confused(int e) const { return e_st(e, ST_CONFUSED) > 0; }
  MSB_HD MSB_INL bool e_frozen(int e) const { return e_st(e, ST_FROZEN) > 0; }

  // ------------------------------------------------------------------------------------------
  // Board primitives
  // ------------------------------------------------------------------------------------------
  // Board.at, board.py:58-65
  MSB_HD MSB_INL int at(P p) const {
    if (p_is_base(p)) return AT_PLAYER + (p.y == 5 ? local() : remote());
    if (!p_valid(p)) return AT_NONE;
    int s = board_at(p_tile(p));
    return s == SLOT_NONE ? AT_NONE : s;
  }
  // Board.set, board.py:67-71
  MSB_HD MSB_INL void board_set(P p, int e) {
    board_put(p_tile(p), e < 0 ? SLOT_NONE : e);
    if (e >= 0) e_set_pos(e, p);
  }

  // A free entity slot: not on the board and not referenced since the step began.  The standard record keeps the
  // set in one 32-bit word; the extended record (up to 128 slots) in four.
  struct Bits {
    static constexpr int NW = NUM_ENT > 128 ? 4 : 2;
    uint64_t w[NW];
    MSB_HD MSB_INL static Bits none() {
      Bits b;
      for (int i = 0; i < NW; i++) b.w[i] = 0;
      return b;
    }
    MSB_HD MSB_INL bool has(int i) const { return ((w[i >> 6] >> (i & 63)) & 1ull) != 0; }
    MSB_HD MSB_INL void add(int i) { w[i >> 6] |= 1ull << (i & 63); }
    MSB_HD MSB_INL void del(int i) { w[i >> 6] &= ~(1ull << (i & 63)); }
    MSB_HD MSB_INL bool any() const {
      uint64_t a = 0;
      for (int i = 0; i < NW; i++) a |= w[i];
      return a != 0;
    }
    MSB_HD MSB_INL int pop() {   // lowest member, removed
      for (int i = 0; i < NW - 1; i++)
        if (w[i]) {
          int b = __builtin_ctzll(w[i]);
          w[i] &= w[i] - 1;
          return 64 * i + b;
        }
      int b = __builtin_ctzll(w[NW - 1]);
      w[NW - 1] &= w[NW - 1] - 1;
      return 64 * (NW - 1) + b;
    }
  };
  MSB_HD MSB_INL static int used_off(int word) { return word == 0 ? H_USED : X_USED_HI + 4 * (word - 1); }
  MSB_HD MSB_INL Bits used_mask() const {
    Bits u = Bits::none();
    u.w[0] = m.ld32(H_USED);
    if (NUM_ENT > 32)
      for (int k = 1; k < USED_WORDS; k++) u.w[k >> 1] |= (uint64_t)m.ld32(used_off(k)) << (32 * (k & 1));
    return u;
  }
  MSB_HD MSB_INL void set_used_mask(Bits u) {
    m.st32(H_USED, (uint32_t)u.w[0]);
    if (NUM_ENT > 32)
      for (int k = 1; k < USED_WORDS; k++) m.st32(used_off(k), (uint32_t)(u.w[k >> 1] >> (32 * (k & 1))));
  }
  MSB_HD MSB_INL int alloc_entity() {
    if (NUM_ENT <= 32) {
      uint32_t used = m.ld32(H_USED);
      uint32_t free_mask = ~used & ((1u << (NUM_ENT & 31)) - 1u);
      if (free_mask == 0) {
        set_fault(FAULT_CAPACITY);
        return 0;
      }
      int e = __builtin_ctz(free_mask);   // lowest free slot (a search loop unrolls into 28 nested exec masks)
      m.st32(H_USED, used | (1u << e));
      return e;
    }
    int e = alloc_entity_ext();
    if (e < 0 && take_back_world_copies()) e = alloc_entity_ext();   // the real game comes first: empty a snapshot
    if (e < 0) {
      set_fault(FAULT_CAPACITY);
      return 0;
    }
    return e;
  }
  MSB_HD MSB_INL int alloc_entity_ext() {   // -1 = no free slot
    for (int w = 0; w < USED_WORDS; w++) {
      const int off = used_off(w);
      uint32_t used = m.ld32(off);
      if (w == USED_WORDS - 1 && (NUM_ENT & 31)) used |= ~0u << (NUM_ENT & 31);   // ids past the last slot are never free
      if (used != 0xffffffffu) {
        int b = __builtin_ctz(~used);
        m.st32(off, used | (1u << b));
        return 32 * w + b;
      }
    }
    return -1;
  }
  // Free the private copies of one frozen world (not the one the engine is acting in); that world becomes partial.
  MSB_HD MSB_NOINLINE bool take_back_world_copies() {
    const int c = ctx();
    for (int w = WORLD_CAP; w >= 1; w--) {
      const int o = world_off(w);
      if (w == c || !m.ld8(o + W_USED)) continue;
      bool any = false;
      Bits u = used_mask();
      for (int t = 0; t < 20; t++) {
        int h = m.ld8(o + W_BOARD + t);
        if (h >= NUM_ENT || m.ld8(E_HOME + h) != (HOME_COPY | w)) continue;
        m.st8(o + W_BOARD + t, SLOT_MISSING);
        u.del(h);
        any = true;
      }
      if (any) {
        set_used_mask(u);
        m.st8(o + W_PARTIAL, 1);
        return true;
      }
    }
    return false;
  }
  // Called at the start of every step: everything not on the board is garbage in the reference
  // (no live Python reference survives a step).
  // the 20 board bytes as five 32-bit rows (row y = tiles 4y..4y+3, x in byte x): one LDS read per row
  MSB_HD MSB_INL uint32_t board_row(int y) const { return m.ld32(OFF_BOARD + 4 * y); }
  // one bit per byte of a board row that is (occupied ? 1 : 0), bit x for tile x
  MSB_HD MSB_INL static uint32_t row_occ4(uint32_t row) {
    uint32_t t = ~row;                                                   // byte == 0 <=> SLOT_NONE (0xFF)
    uint32_t nz = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;  // bit 7 of every non-zero byte
    return (((nz >> 7) * 0x00204081u) >> 21) & 0xFu;                     // gather bits 0,8,16,24 -> 0..3
  }
  // occupied / empty tiles as 20-bit masks (bit y*4+x): five LDS reads, no per-tile loop
  MSB_HD MSB_INL uint32_t occ_mask() const {
    uint32_t o = 0;
    for (int y = 0; y < 5; y++) o |= row_occ4(board_row(y)) << (4 * y);
    return o;
  }
  MSB_HD MSB_INL uint32_t empty_mask() const { return ~occ_mask() & 0xFFFFFu; }
  MSB_HD MSB_INL void begin_step() {
    MSB_SCOPE(PS_BEGIN_STEP);
    // one shift per tile, no compare: an empty tile (0xFF) sets the top bit, which is not an entity slot
    static_assert(NUM_ENT > 32 || (SLOT_NONE & 31) >= NUM_ENT, "the empty marker must map outside the slot bits");
    Bits used = Bits::none();
    for (int y = 0; y < 5; y++) {
      uint32_t row = board_row(y);
      if (NUM_ENT <= 32) {
        for (int x = 0; x < 4; x++) used.w[0] |= 1u << ((row >> (8 * x)) & 31u);
      } else {
        for (int x = 0; x < 4; x++) {
          uint32_t sl = (row >> (8 * x)) & 0xffu;
          if (sl < (uint32_t)NUM_ENT) used.add((int)sl);
        }
      }
    }
    if (NUM_ENT <= 32) used.w[0] &= (1ull << (NUM_ENT & 31)) - 1ull;
    // a hand/deck entry aliasing an entity that has left the board keeps that object's last strength
    if (m.ld8(H_OBSFAULT) & GF_ALIAS) {
      for (int o = 0; o < 2; o++) {
        int hn = pl_hand_n(o), dn = pl_deck_n(o);
        for (int i = 0; i < hn + dn; i++) {
          int off = i < hn ? hand_ref(o, i) : deck_ref(o, i - hn);
          int fl = m.ld8(off + 2);
          if (!(fl & CF_ALIAS)) continue;
          int slot = m.ld8(off + 3);
          if (used.has(slot)) continue;
          int str = e_str(slot);
          if (str < 0 || str > 255) {
            set_fault(FAULT_CAP_INST);
            str = 0;
          }
          m.st8(off + 2, (fl & ~CF_ALIAS) | CF_STR);
          m.st8(off + 3, str);
#if defined(MSB_EXT) && MSB_EXT
          m.st8(pl(o, P_IPOS + (off - pl(o, P_INST)) / 4), m.ld8g(eg(slot), EO_POS));   // ... and its last position
#endif
        }
      }
    }
    if (REM_LISTS) used = rem_collect(used);
    set_used_mask(used);
    m.st8(H_DEPTH, 0);
  }
  MSB_HD MSB_INL int new_entity(int card, int owner, int strength, int movement, bool ff) {
    int r_ = new_entity_impl(card, owner, strength, movement, ff);
    return r_;
  }
  MSB_HD MSB_A_NEWENT int new_entity_impl(int card, int owner, int strength, int movement, bool ff) {
    int e = alloc_entity();
    if (fault()) return e;
    // card | flags<<8 | pos<<16 | mov<<24 ; st0..st3 = 0 ; st4 = 0, move_id = 0, strength<<16 ; dmg = 0, path_n = 0
    msb_u32x4 g = {(uint32_t)card | ((uint32_t)((owner ? EF_OWNER : 0) | (ff ? EF_FF : 0)) << 8) | ((uint32_t)(movement & 0xff) << 24),
                   0u, (uint32_t)(strength & 0xffff) << 16, (uint32_t)kind_byte(card) << 24};
    m.st128g(eg(e), g);
    e_set_path(e, 0);
    if (REM_LISTS) {
      m.st8(E_REM + e, REM_NONE);
      m.st8(E_HOME + e, ctx());   // token.player = <a player of the world the spawning entity lives in>
    }
    return e;
  }

  // ------------------------------------------------------------------------------------------
  // Worlds (extended record only; state.h).  ctx = the world whose board / players / trigger stack / stream the
  // record's own fields currently hold: an entity method of the reference reaches them through self.player, so it
  // runs in the world of the entity it is called on (ctx_enter / ctx_leave around every such method).
  // ------------------------------------------------------------------------------------------
  MSB_HD MSB_INL int ctx() const { return REM_LISTS ? m.ld8(X_CTX) : 0; }
  MSB_HD MSB_INL int e_home(int e) const { return REM_LISTS ? (m.ld8(E_HOME + e) & 0x7f) : 0; }
  MSB_HD MSB_INL static int world_off(int w) { return OFF_WORLD + (w - 1) * WORLD_BYTES; }
  MSB_HD MSB_INL void swap8(int a, int b) {
    int x = m.ld8(a), y = m.ld8(b);
    m.st8(a, y);
    m.st8(b, x);
  }
  MSB_HD MSB_INL void swap32(int a, int b) {
    uint32_t x = m.ld32(a), y = m.ld32(b);
    m.st32(a, y);
    m.st32(b, x);
  }
  // exchange the record's own world fields with storage w
  MSB_HD MSB_NOINLINE void swap_world(int w) {
    const int o = world_off(w);
    static_assert(OFF_BOARD % 4 == 0 && OFF_TRIG % 4 == 0 && OFF_WORLD % 4 == 0, "word-wise swaps");
    for (int k = 0; k < 5; k++) swap32(OFF_BOARD + 4 * k, o + W_BOARD + 4 * k);
    for (int k = 0; k < 5; k++) swap32(OFF_TRIG + 4 * k, o + W_TRIG + 4 * k);
    if (TRIG_WIDE) swap32(X_TRIGSRC, o + W_TRIGSRC);
    swap8(H_TOPLAY, o + W_TOPLAY);
    swap8(H_TRIG_N, o + W_TRIG_N);
    swap8(H_RESOLVING, o + W_RESOLVING);
    swap8(H_PHASE, o + W_PHASE);
    swap8(H_CP, o + W_CP);
    for (int p = 0; p < 2; p++) {
      swap8(pl(p, P_FRONT), o + W_FRONT + p);
      int x = m.ld16(pl(p, P_BASE)), y = m.ld16(o + W_BASE + 2 * p);
      m.st16(pl(p, P_BASE), y);
      m.st16(o + W_BASE + 2 * p, x);
    }
  }
  // Invariant: with ctx = c != 0 the record's fields hold world c and storage c holds the real game's.
  MSB_HD MSB_INL void switch_ctx(int w) {
    int c = ctx();
    if (c == w) return;
    if (w == WORLD_LOST || (w != 0 && m.ld8(world_off(w) + W_PARTIAL))) {
      set_fault(FAULT_CAPACITY);   // the snapshot this entity lives in could not be kept (state.h: slots are a cache)
      return;
    }
    if (c) swap_world(c);
    if (w) swap_world(w);
    m.st8(X_CTX, w);
  }
  // enter the world of entity e; returns the world to come back to (-1 = no switch happened)
  MSB_HD MSB_INL int ctx_enter(int e) {
    if (!REM_LISTS) return -1;
    int h = e_home(e), c = ctx();
    if (h == c) return -1;
    switch_ctx(h);
    return c;
  }
  MSB_HD MSB_INL void ctx_leave(int saved) {
    if (REM_LISTS && saved >= 0 && !fault()) switch_ctx(saved);
  }
  // one draw of the current world's stream copy: an absolute position in the game's stream
  MSB_HD MSB_NOINLINE uint32_t world_rng_next() {
    const int o = world_off(ctx());
    uint32_t abs = m.ld32(o + W_RNG);
    uint32_t blk = abs / (uint32_t)MT_N, idx = abs % (uint32_t)MT_N, cur = (uint32_t)m.ld16(X_RNGBLK) & 0xffffu;
    m.st32(o + W_RNG, abs + 1);
    uint64_t p = 0;
    if (blk == cur) p = m.ld64(H_RNGCUR);
    else if (blk == cur + 1) p = m.ld64(H_RNGNXT);
    else if (blk < cur) return mt_regen_word(m.ld32(X_SEED), blk, idx);   // that block is no longer resident
    if (p == 0) {
      m.st8(H_RNGOVER, 1);
      return 0;
    }
    return ((MSB_RNG_PTR)(uintptr_t)p)[idx];
  }
  // output idx of block blk of RandomState(seed), from scratch (rare: a world older than the two resident blocks)
  MSB_HD MSB_NOINLINE static uint32_t mt_regen_word(uint32_t seed, uint32_t blk, uint32_t idx) {
    uint32_t mt[MT_N];
    mt_seed(mt, seed);
    for (uint32_t b = 0; b <= blk; b++) mt_twist(mt);
    return mt_temper(mt[idx]);
  }
  // the record's stream window moved on by one block (called by whoever commits the cursor)
  MSB_HD MSB_INL void rng_block_advance() {
    if (REM_LISTS) m.st16(X_RNGBLK, (int)(((uint32_t)m.ld16(X_RNGBLK) + 1u) & 0xffffu));
  }
  MSB_HD MSB_INL void set_seed(uint32_t seed) {
    if (REM_LISTS) m.st32(X_SEED, seed);
  }

  // ---- b005's remembered copies (extended record only) -----------------------------------------
  MSB_HD MSB_INL static int rem_off(int list) { return OFF_REM + list * REM_LIST_BYTES; }
  MSB_HD MSB_INL int rem_n(int list) const { return m.ld8(rem_off(list)); }
  MSB_HD MSB_INL int rem_get(int list, int i) const { return m.ld8(rem_off(list) + 4 + i); }
  MSB_HD MSB_INL int rem_alloc() {
    for (int l = 0; l < REM_LISTS; l++)
      if (m.ld8(rem_off(l) + 1) == 0) {
        m.st8(rem_off(l), 0);
        m.st8(rem_off(l) + 1, 1);
        return l;
      }
    set_fault(FAULT_CAP_REM);
    return REM_NONE;
  }
  MSB_HD MSB_INL int rem_alloc_soft() {   // REM_LOST instead of a fault
    for (int l = 0; l < REM_LISTS; l++)
      if (m.ld8(rem_off(l) + 1) == 0) {
        m.st8(rem_off(l), 0);
        m.st8(rem_off(l) + 1, 1);
        return l;
      }
    return REM_LOST;
  }
  MSB_HD MSB_INL int world_alloc() {
    for (int w = 1; w <= WORLD_CAP; w++)
      if (m.ld8(world_off(w) + W_USED) == 0) {
        m.st8(world_off(w) + W_USED, 1);
        return w;
      }
    return WORLD_LOST;   // no storage: the entities that would live in it are lost to the game unless they never act
  }
  // Garbage collection at the start of a step: what the real board reaches survives -- entities on it, the memory
  // lists of those (and, nested, of the remembered copies), the worlds such entities belong to and everything on a
  // live world's board or trigger stack.  Returns the set of live entity slots.
  MSB_HD MSB_NOINLINE Bits rem_collect(Bits used) {
    Bits todo = used;
    uint64_t lists = 0;
    uint32_t worlds = 0;
    while (todo.any()) {
      const int e = todo.pop();
      const int L = m.ld8(E_REM + e);
      if (L != REM_NONE && L < REM_LISTS && !((lists >> L) & 1ull)) {
        lists |= 1ull << L;
        int n = rem_n(L);
        for (int k = 0; k < n; k++) {
          int r = rem_get(L, k);
          if (r < NUM_ENT && !used.has(r)) {
            used.add(r);
            todo.add(r);
          }
        }
      }
      const int hb = m.ld8(E_HOME + e);
      const int H = (hb & HOME_COPY) ? 0 : hb;   // a snapshot's private copy keeps nothing alive by itself
      if (H != 0 && H <= WORLD_CAP && !((worlds >> H) & 1u)) {
        worlds |= 1u << H;
        const int o = world_off(H);
        int tn = m.ld8(o + W_TRIG_N);
        for (int t = 0; t < 20 + tn; t++) {
          int v = t < 20 ? m.ld8(o + W_BOARD + t) : (m.ld8(o + W_TRIG + t - 20) & TRIG_SLOT);
          if (v >= NUM_ENT) continue;   // SLOT_NONE / SLOT_MISSING
          if (!used.has(v)) {
            used.add(v);
            todo.add(v);
          }
        }
      }
    }
    for (int l = 0; l < REM_LISTS; l++) m.st8(rem_off(l) + 1, (int)((lists >> l) & 1ull));
    for (int w = 1; w <= WORLD_CAP; w++) m.st8(world_off(w) + W_USED, (worlds >> w) & 1u);
    return used;
  }
  // a new slot with the same attributes as h (path included), no memory of its own, same world
  MSB_HD MSB_INL int dup_entity(int h, bool may_fail = false) {
    int c = may_fail ? alloc_entity_ext() : alloc_entity();
    if (c < 0 || fault()) return c;
    m.st128g(eg(c), m.ld128g(eg(h)));
    e_set_path(c, e_path(h));
    m.st8(E_REM + c, REM_NONE);
    m.st8(E_HOME + c, m.ld8(E_HOME + h) & 0x7f);
    return c;
  }
  // deepcopy of world src as it is now (from any context: a restored b005 remembers inside its own world).  The entity the
  // running deepcopy started from keeps its identity inside the snapshot: where src's board holds root_old the
  // snapshot holds root_new (deepcopy's memo).
  MSB_HD MSB_NOINLINE int world_snapshot(int src, int root_old, int root_new) {
    if (src == WORLD_LOST) return WORLD_LOST;
    const int w = world_alloc();
    if (w == WORLD_LOST) return w;
    // Where world src's swapped fields are right now (swap_world): in the record's own fields if the engine is acting
    // in it ("live"), else in a storage slot -- its own, or, for the real game while ctx = c != 0, slot c.  W_PARTIAL
    // and W_RNG never move: they stay in the world's own slot (the real game: never partial, the record's cursor).
    const int c = ctx();
    const bool live = src == c;
    const int o = world_off(w), so = live ? 0 : world_off(src ? src : c);
    m.st8(o + W_PARTIAL, src ? m.ld8(world_off(src) + W_PARTIAL) : 0);
    m.st8(o + W_TOPLAY, !live ? m.ld8(so + W_TOPLAY) : m.ld8(H_TOPLAY));
    m.st8(o + W_RESOLVING, !live ? m.ld8(so + W_RESOLVING) : m.ld8(H_RESOLVING));
    m.st8(o + W_PHASE, !live ? m.ld8(so + W_PHASE) : m.ld8(H_PHASE));
    m.st8(o + W_CP, !live ? m.ld8(so + W_CP) : m.ld8(H_CP));
    for (int p = 0; p < 2; p++) {
      m.st8(o + W_FRONT + p, !live ? m.ld8(so + W_FRONT + p) : m.ld8(pl(p, P_FRONT)));
      m.st16(o + W_BASE + 2 * p, !live ? m.ld16(so + W_BASE + 2 * p) : m.ld16(pl(p, P_BASE)));
    }
    m.st32(o + W_RNG, src ? m.ld32(world_off(src) + W_RNG)
                          : ((uint32_t)m.ld16(X_RNGBLK) & 0xffffu) * (uint32_t)MT_N + ((uint32_t)m.ld16(H_RNGPOS) & 0xffffu));
    for (int t = 0; t < 20; t++) {
      int h = !live ? m.ld8(so + W_BOARD + t) : board_at(t);
      int nh = h;   // SLOT_NONE and SLOT_MISSING carry over
      if (h < NUM_ENT) {
        if (h == root_old) {
          nh = root_new;
        } else if (m.ld8(o + W_PARTIAL)) {
          nh = SLOT_MISSING;
        } else {
          nh = dup_entity(h, true);
          if (nh < 0) {   // out of slots: the snapshot stays partial (state.h: slots are a cache)
            nh = SLOT_MISSING;
            m.st8(o + W_PARTIAL, 1);
          } else {
            m.st8(E_HOME + nh, HOME_COPY | w);   // its .player is the snapshot's player; it exists on this board only
          }
        }
      }
      m.st8(o + W_BOARD + t, nh);
    }
    // pending triggers name entity objects: the copies standing on the same tiles (anything else is out of reach)
    int tn = !live ? m.ld8(so + W_TRIG_N) : m.ld8(H_TRIG_N);
    m.st8(o + W_TRIG_N, tn);
    for (int i = 0; i < tn; i++) {
      int v = !live ? m.ld8(so + W_TRIG + i) : m.ld8(OFF_TRIG + i);
      int nv = -1;
      for (int t = 0; t < 20; t++)
        if ((!live ? m.ld8(so + W_BOARD + t) : board_at(t)) == (v & TRIG_SLOT)) nv = m.ld8(o + W_BOARD + t);
      if (nv < 0 || nv >= NUM_ENT) {
        m.st8(o + W_PARTIAL, 1);
        nv = 0;
      }
      m.st8(o + W_TRIG + i, nv | (TRIG_WIDE ? 0 : (v & 0x80)));
    }
    if (TRIG_WIDE) m.st32(o + W_TRIGSRC, !live ? m.ld32(so + W_TRIGSRC) : m.ld32(X_TRIGSRC));
    return w;
  }
  // copy.deepcopy of a memory list (the list object, its entities, their own memories, and -- through entity.player
  // -- the world each of them belongs to, once per deepcopy call: memo[]).  The reference's recursion over nested
  // memories is an explicit depth-first walk here (pre-order, so lists, slots and worlds are handed out in the same
  // order): level L copies list s_src[L] into s_dst[L], s_k[L] is its next element and s_c[L] the copy whose own memory
  // the level below is making.
  MSB_HD MSB_NOINLINE int rem_deep_copy(int src0, int* memo, int root_old, int root_new) {
    int s_src[REM_DEPTH + 2], s_dst[REM_DEPTH + 2], s_k[REM_DEPTH + 2], s_c[REM_DEPTH + 2];
    int L = 0, ret = REM_NONE;
    enum { CALL, LOOP, RET } mode = CALL;
    s_src[0] = src0;
    for (;;) {
      if (mode == CALL) {   // deepcopy of list s_src[L] begins
        const int src = s_src[L];
        if (L > REM_DEPTH) {
          set_fault(FAULT_CAP_REM);
          ret = REM_NONE;
          mode = RET;
          continue;
        }
        if (src == REM_LOST) {
          ret = REM_LOST;
          mode = RET;
          continue;
        }
        const int dst = rem_alloc_soft();
        if (dst == REM_LOST) {   // no list storage: the copy's memory is lost unless it is never used
          ret = REM_LOST;
          mode = RET;
          continue;
        }
        m.st8(rem_off(dst), rem_n(src));
        s_dst[L] = dst;
        s_k[L] = 0;
        mode = LOOP;
      } else if (mode == LOOP) {   // next element of the list being copied at level L
        const int src = s_src[L], dst = s_dst[L], k = s_k[L];
        if (k >= rem_n(src) || fault()) {
          ret = dst;
          mode = RET;
          continue;
        }
        const int r = rem_get(src, k);
        const int c = dup_entity(r);
        if (fault()) {
          ret = dst;
          mode = RET;
          continue;
        }
        m.st8(rem_off(dst) + 4 + k, c);
        const int H = m.ld8(E_HOME + r) & 0x7f;
        int nh = WORLD_LOST;
        if (H != WORLD_LOST) {
          if (memo[H] < 0) memo[H] = world_snapshot(H, root_old, root_new);
          if (fault()) {
            ret = dst;
            mode = RET;
            continue;
          }
          nh = memo[H];
        }
        m.st8(E_HOME + c, nh);
        s_k[L] = k + 1;
        const int L2 = m.ld8(E_REM + r);
        if (L2 == REM_LOST || (L2 != REM_NONE && rem_n(L2) > 0)) {
          s_c[L] = c;
          L++;
          s_src[L] = L2;
          mode = CALL;
        }
      } else {   // level L is done: `ret` is its copy
        if (L == 0) return ret;
        L--;
        if (fault()) {   // the level above breaks out of its loop and hands back what it has
          ret = s_dst[L];
          continue;
        }
        m.st8(E_REM + s_c[L], ret);
        mode = LOOP;
      }
    }
  }
  // Card.copy() of the on-board entity e (card.py:71-75): deepcopy, then copied.player = self.player
  MSB_HD MSB_NOINLINE int rem_copy_entity(int e) {
    int c = dup_entity(e);   // same world as e: the one attribute that is re-bound
    if (fault()) return c;
    int L = m.ld8(E_REM + e);
    if (L == REM_LOST) {
      m.st8(E_REM + c, REM_LOST);
    } else if (L != REM_NONE && rem_n(L) > 0) {
      int memo[WORLD_CAP + 1];
      for (int i = 0; i <= WORLD_CAP; i++) memo[i] = -1;
      int d = rem_deep_copy(L, memo, e, c);
      if (fault()) return c;
      m.st8(E_REM + c, d);
    }
    return c;
  }


  // Board.calculate_front_line, board.py:78-92.  `player` is an order; the reference compares
  // Player objects by order (player.py:39-40).
  MSB_HD MSB_A_FRONT void calculate_front_line(int player) {
    // any(board[y][x] is not None and board[y][x].player == player for x in range(4)) per row: the first
    // (local: lowest y, remote: highest y) row holding one of the player's entities; only occupied tiles are read
    const uint32_t r0 = board_row(0), r1 = board_row(1), r2 = board_row(2), r3 = board_row(3), r4 = board_row(4);
    uint32_t occ = row_occ4(r0) | (row_occ4(r1) << 4) | (row_occ4(r2) << 8) | (row_occ4(r3) << 12) | (row_occ4(r4) << 16);
    const unsigned long long b01 = (unsigned long long)r0 | ((unsigned long long)r1 << 32);
    const unsigned long long b23 = (unsigned long long)r2 | ((unsigned long long)r3 << 32);
    const bool is_local = player == local();
    int found = -1;
    while (occ) {
      const int t = is_local ? __builtin_ctz(occ) : 31 - __builtin_clz(occ);
      occ &= ~(1u << t);
      const unsigned long long bw = t < 8 ? b01 : (t < 16 ? b23 : (unsigned long long)r4);
      const int s = (int)((bw >> (8 * (t & 7))) & 0xff);
      if (e_owner(s) == player) {
        found = t >> 2;
        break;
      }
    }
    if (is_local)
      set_pl_front(local(), found < 0 ? 4 : (found > 1 ? found : 1));
    else
      set_pl_front(remote(), found < 0 ? 0 : (found < 3 ? found : 3));
  }
  // board.calculate_front_line(board.current_player.opponent), unit.py:231 / structure.py:69.
  // Player.opponent (player.py:42-44) is board.remote for the FIRST player, board.local for the
  // SECOND -- i.e. the second player's "opponent" is itself while it is the mover (fact #3).
  MSB_HD MSB_INL int opponent_of(int order) const { return order == 0 ? remote() : local(); }
  MSB_HD MSB_INL void recalc_front_after_destroy() { calculate_front_line(opponent_of(cp())); }

  // Board.get_targets, board.py:147-204
  MSB_HD MSB_INL PList get_targets(int pov, Tgt t, int exclude_pk) {
    PList r_ = get_targets_impl(pov, t, exclude_pk);
    return r_;
  }
  MSB_HD MSB_A_TARGETS PList get_targets_impl(int pov, Tgt t, int exclude_pk) {
    MSB_SCOPE(PS_GET_TARGETS);
    PList out;
    out.clear();
    const bool asc = (pov == local());
    const int kind = tg_kind(t), side = tg_side(t), limit = tg_limit(t);
    const int types = tg_types(t), xtypes = tg_xtypes(t), status = tg_status(t), xstatus = tg_xstatus(t);
    const bool want_types = (types | xtypes | (tg_non_hero(t) ? 1 : 0)) != 0;
    // Pass 1: visit only the occupied tiles and collect the matching ones as a tile bit mask (the board is
    // read as five 32-bit rows; one 16-byte LDS read per entity).  Pass 2 emits them in iteration order.
    const uint32_t r0 = board_row(0), r1 = board_row(1), r2 = board_row(2), r3 = board_row(3), r4 = board_row(4);
    uint32_t occ = row_occ4(r0) | (row_occ4(r1) << 4) | (row_occ4(r2) << 8) | (row_occ4(r3) << 12) | (row_occ4(r4) << 16);
    const unsigned long long b01 = (unsigned long long)r0 | ((unsigned long long)r1 << 32);
    const unsigned long long b23 = (unsigned long long)r2 | ((unsigned long long)r3 << 32);
    uint32_t hit = 0;
    while (occ) {
      const int tile = __builtin_ctz(occ);
      occ &= occ - 1;
      const unsigned long long bw = tile < 8 ? b01 : (tile < 16 ? b23 : (unsigned long long)r4);
      const int e = (int)((bw >> (8 * (tile & 7))) & 0xff);
      const msb_u32x4 g = m.ld128g(eg(e));   // the whole entity in one LDS read
      int str = (int)(int16_t)(g[2] >> 16);
      if (str <= 0) continue;
      int c = (int)(g[0] & 0xff);
      bool is_unit = ((g[3] >> 24) & EK_UNIT) != 0;
      bool strength_ok = limit == LIMIT_NONE || str <= limit;
      bool ok;
      if (is_unit) {
        int ty = want_types ? card_types(c) : 0;   // table lookup only for type filters
        bool type_ok = types == 0 || (ty & types) != 0;
        bool xtype_ok = xtypes == 0 || (ty & xtypes) == 0;
        bool hero_ok = !tg_non_hero(t) || !(ty & (1 << UT_HERO));
        int stm = 0;
        if (status | xstatus) {
          for (int s = 0; s < 4; s++)
            if ((g[1] >> (8 * s)) & 0xff) stm |= 1 << s;
          if (g[2] & 0xff) stm |= 1 << 4;
        }
        bool st_ok = status == 0 || (stm & status) != 0;
        bool xst_ok = xstatus == 0 || (stm & xstatus) == 0;
        ok = type_ok && xtype_ok && hero_ok && st_ok && xst_ok && strength_ok && (kind == TK_ANY || kind == TK_UNIT);
      } else {
        ok = strength_ok && (kind == TK_ANY || kind == TK_STRUCTURE);
      }
      int own = (int)((g[0] >> 8) & EF_OWNER);
      bool side_ok = side == TS_ANY || (side == TS_FRIENDLY && own == pov) || (side == TS_ENEMY && own != pov);
      if (ok && side_ok) hit |= 1u << tile;
    }
    // ascending tiles for the local point of view, descending for the remote one (board.py:160-166)
    {
      unsigned long long w0 = 0, w1 = 0;
      int n = 0;
      while (hit) {
        const int tile = asc ? __builtin_ctz(hit) : 31 - __builtin_clz(hit);
        hit &= ~(1u << tile);
        const unsigned long long v = (unsigned long long)((((tile >> 2) + 1) << 3) | ((tile & 3) + 1));   // p_pack(tile_p(tile))
        if (n < 10) w0 |= v << (6 * n);
        else w1 |= v << (6 * (n - 10));
        n++;
      }
      out.w = msb_u64x4{w0, w1, 0ull, (unsigned long long)n};
    }
    if (tg_base(t)) {
      P friendly = asc ? P{-1, 5} : P{-1, -1};
      P enemy = asc ? P{-1, -1} : P{-1, 5};
      if (side == TS_FRIENDLY || side == TS_ANY) out.push(friendly);
      if (side == TS_ENEMY || side == TS_ANY) out.push(enemy);
    }
    if (exclude_pk != PK_NONE) {
      int m = out.n();
      for (int i = 0; i < m; i++)
        if (out.get(i) == exclude_pk) {
          out.remove_at(i);
          break;
        }
    }
    return out;
  }

  // membership of `p` in the raw tile list of a geometric selector (board.py:206-296)
  MSB_HD MSB_INL bool in_shape(int shape, P c, int pov, P p) const {
    int dx = p.x - c.x, dy = p.y - c.y;
    switch (shape) {
      case SH_FRONT: return dx == 0 && (pov == local() ? (p.y < c.y && p.y >= 0) : (p.y > c.y && p.y <= 4));
      case SH_BEHIND: return dx == 0 && (pov == local() ? (p.y > c.y && p.y <= 4) : (p.y < c.y && p.y >= 0));
      case SH_SIDE: return dy == 0 && (dx == 1 || dx == -1);
      case SH_ROW: return dy == 0 && p.x >= 0 && p.x < 4;
      case SH_COLUMN: return dx == 0 && p.y >= 0 && p.y < 5;
      case SH_BORDERING: return (dy == 0 && (dx == 1 || dx == -1)) || (dx == 0 && (dy == 1 || dy == -1));
      default: return !(dx == 0 && dy == 0) && dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1;
    }
  }
  // get_front_tiles .. get_surrounding_tiles WITHOUT a target: fixed enumeration order.
  MSB_HD MSB_A_TILES PList shape_tiles(int shape, P c, int pov) {
    PList out;
    out.clear();
    switch (shape) {
      case SH_FRONT:
        if (pov == local()) {
          for (int y = c.y - 1; y >= 0; y--) out.push(P{c.x, y});
        } else {
          for (int y = c.y + 1; y < 5; y++) out.push(P{c.x, y});
        }
        break;  // the trailing sort (board.py:217) keeps this order
      case SH_BEHIND:
        if (pov == local()) {
          for (int y = c.y + 1; y < 5; y++) out.push(P{c.x, y});
        } else {
          for (int y = c.y - 1; y >= 0; y--) out.push(P{c.x, y});
        }
        break;
      case SH_SIDE: {
        P a{c.x - 1, c.y}, b{c.x + 1, c.y};
        if (p_valid(a)) out.push(a);
        if (p_valid(b)) out.push(b);
      } break;
      case SH_ROW:
        for (int x = 0; x < 4; x++) {
          P a{x, c.y};
          if (p_valid(a)) out.push(a);
        }
        break;
      case SH_COLUMN:
        for (int y = 0; y < 5; y++) {
          P a{c.x, y};
          if (p_valid(a)) out.push(a);
        }
        break;
      case SH_BORDERING: {
        const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
        for (int i = 0; i < 4; i++) {
          P a{c.x + dx[i], c.y + dy[i]};
          if (p_valid(a)) out.push(a);
        }
      } break;
      default: {
        const int dx[8] = {-1, -1, -1, 1, 1, 1, 0, 0}, dy[8] = {0, -1, 1, 0, -1, 1, -1, 1};
        for (int i = 0; i < 8; i++) {
          P a{c.x + dx[i], c.y + dy[i]};
          if (p_valid(a)) out.push(a);
        }
      } break;
    }
    return out;
  }
  // ... WITH a target: get_targets order filtered by membership; front/behind re-sorted by y.
  MSB_HD MSB_A_SHAPE PList shape_targets(int shape, P c, int pov, Tgt t, int exclude_pk) {
    PList all = get_targets(pov, t, exclude_pk);
    PList out;
    out.clear();
    bool base_rule = (shape == SH_FRONT || shape == SH_BEHIND || shape == SH_COLUMN || shape == SH_BORDERING ||
                      shape == SH_SURROUNDING);
    int m = all.n();
    for (int i = 0; i < m; i++) {
      P p = all.at(i);
      if (in_shape(shape, c, pov, p) || (base_rule && p_is_base(p) && tg_base(t))) out.push_raw(all.get(i));
    }
    if (shape == SH_FRONT || shape == SH_BEHIND) {
      // stable sort by y; reverse=True keeps equal keys in original order too (CPython list.sort)
      bool desc = (shape == SH_FRONT) ? (pov == local()) : (pov == remote());
      int on = out.n();
      for (int i = 1; i < on; i++) {
        int k = out.get(i);
        int ky = p_unpack(k).y;
        int j = i - 1;
        while (j >= 0 && (desc ? p_unpack(out.get(j)).y < ky : p_unpack(out.get(j)).y > ky)) {
          out.set(j + 1, out.get(j));
          j--;
        }
        out.set(j + 1, k);
      }
    }
    return out;
  }

  // numpy RandomState.choice(list) / shuffle(list)
  MSB_HD MSB_INL int choice_index(int n) { return rng_randint(0, n); }
  MSB_HD MSB_INL P choice_point(PList l) { return l.at(choice_index(l.n())); }
  MSB_HD MSB_A_SHUFFLE PList shuffle(PList l) {
    for (int i = l.n() - 1; i >= 1; i--) {
      int j = (int)rng_interval((uint32_t)i);
      int tmp = l.get(i);
      l.set(i, l.get(j));
      l.set(j, tmp);
    }
    return l;
  }
  // targets.sort(key=lambda t: (key(t), random.random()), reverse=rev)[:k] for k <= 2, the only uses the
  // cards make of it (b002 b008 b009 b104 s101).  random() is called once per element in list order
  // BEFORE sorting; the sort is stable and reverse keeps ties in order, so the first k of the sorted
  // list are the k best under (key, r) with earlier elements winning ties.  key_mode: 0 = y, 1 = strength.
  MSB_HD MSB_A_SHUFFLE PList sorted_head(PList l, int key_mode, bool rev, int k) {
    int b0 = -1, b1 = -1, k0 = 0, k1 = 0;
    double r0 = 0.0, r1 = 0.0;
    int m = l.n();
    for (int i = 0; i < m; i++) {
      P p = l.at(i);
      int kv = key_mode == 0 ? p.y : e_str(at(p));
      double rv = rng_random_sample();
      // does element i sort strictly before the current best / second best?
      bool lt0 = b0 < 0 || (rev ? (kv > k0 || (kv == k0 && rv > r0)) : (kv < k0 || (kv == k0 && rv < r0)));
      if (lt0) {
        b1 = b0; k1 = k0; r1 = r0;
        b0 = i; k0 = kv; r0 = rv;
      } else {
        bool lt1 = b1 < 0 || (rev ? (kv > k1 || (kv == k1 && rv > r1)) : (kv < k1 || (kv == k1 && rv < r1)));
        if (lt1) {
          b1 = i; k1 = kv; r1 = rv;
        }
      }
    }
    PList out;
    out.clear();
    if (b0 >= 0 && k >= 1) out.push_raw(l.get(b0));
    if (b1 >= 0 && k >= 2) out.push_raw(l.get(b1));
    return out;
  }

  // ------------------------------------------------------------------------------------------
  // Control flow: an explicit work stack instead of the reference's recursion
  // ------------------------------------------------------------------------------------------
  // The reference recurses: Unit.move -> activate_ability -> deal_damage -> destroy -> pop_trigger -> activate_ability
  // -> command -> move ... (unit.py:124-231, card.py:48-62, board.py:46-56).  Here every function on such a cycle is a
  // FRAME on the game's work stack (M::sk_ld / sk_st: on the device LDS words next to the record, state.h) and run()
  // is the only loop: it takes the top frame and executes its handler until the handler "calls" -- pushes the callee's
  // frame and returns to run() -- or finishes and pops itself.  A handler is a forward-only state machine: the
  // frame's `state` says where to resume, and a backward jump (the next round of a loop whose body calls) goes through
  // run().  A call that completes without pushing anything (damage that kills nobody and fires nothing) continues
  // inline.  A fault ends the step at once, like the exception it stands for (run() drops the stack).
  //   frame = header word on top {fn, state, a, b: one byte each} + the frame's other words below it
  //   call_X(k, ...): the part of X before its first nested call runs at once, in the caller; what is left of X, if
  //   anything, waits in a frame.  The caller tells "completed" from "pending" by the stack pointer.
  struct Wk {
    int sp;       // words in use on this game's work stack (the resident part)
    int result;   // Stormbound.step's reward | done << 1
    int base;     // words evicted so far (wk_evict; always 0 on the host)
    int seg;      // evictions pending
  };
  enum : int { F_MOVE = 1, F_RUNAB, F_CTXLEAVE, F_DESTROY_TAIL, F_CMD_TAIL, F_EACH, F_AFTER, F_TURN, F_EVICTED };
  // Words of a frame, by function (the eviction below moves whole frames).
  MSB_HD MSB_INL static int frame_words(int fn) {
    return fn == F_MOVE ? 4 : fn == F_RUNAB ? 2 : fn == F_EACH ? 8 : fn == F_AFTER ? 2 : fn == F_TURN ? 7 : 1;
  }
  // Where the resident part of the stack (M::SKW words of LDS on the device) could run short during the next handler --
  // one handler pushes at most SK_NEED words before it returns to run() -- everything below the top frame moves out to
  // the eviction block, the top frame slides down and an F_EVICTED frame {hdr: n} marks the place: when that frame is
  // on top again, the n words come back.  Handlers never notice.
  MSB_HD MSB_INL void wk_evict(Wk& k, int fn) {
    const int fs = frame_words(fn), n = k.sp - fs;
    for (int i = 0; i < n; i++) M::ovf_st(k.base + i, m.sk_ld(i));
    for (int j = 0; j < fs; j++) m.sk_st(1 + j, m.sk_ld(n + j));
    m.sk_st(0, mk_hdr(F_EVICTED, 0, n, 0));
    k.base += n;
    k.seg++;
    k.sp = 1 + fs;
  }
  MSB_HD MSB_INL void h_evicted(Wk& k, const uint32_t hdr) {
    const int n = hdr_a(hdr);
    k.base -= n;
    k.seg--;
    for (int i = 0; i < n; i++) m.sk_st(i, M::ovf_ld(k.base + i));
    k.sp = n;
  }
  // Called where a step enters an ability or a move: false = the stack is deeper than any chain the depth limit allows.
  MSB_HD MSB_INL bool wk_reserve(const Wk& k) const { return k.base + k.sp - k.seg <= SK_CAP - SK_MARGIN; }
  MSB_HD MSB_INL static uint32_t mk_hdr(int fn, int st, int a, int b) {
    return (uint32_t)(fn & 0xff) | ((uint32_t)(st & 0xff) << 8) | ((uint32_t)(a & 0xff) << 16) | ((uint32_t)(b & 0xff) << 24);
  }
  MSB_HD MSB_INL static int hdr_fn(uint32_t h) { return (int)(h & 0xff); }
  MSB_HD MSB_INL static int hdr_st(uint32_t h) { return (int)((h >> 8) & 0xff); }
  MSB_HD MSB_INL static int hdr_a(uint32_t h) { return (int)((h >> 16) & 0xff); }
  MSB_HD MSB_INL static int hdr_b(uint32_t h) { return (int)(h >> 24); }
  MSB_HD MSB_INL void wk_push(Wk& k, uint32_t v) {
    m.sk_st(k.sp, v);
    k.sp++;
  }
  // a point / slot list as six stack words (the three 64-bit lanes of a PList; the length travels in the frame)
  MSB_HD MSB_INL void wk_push_list(Wk& k, const PList& l) {
    for (int j = 2; j >= 0; j--) {
      wk_push(k, (uint32_t)(l.w[j] >> 32));
      wk_push(k, (uint32_t)l.w[j]);
    }
  }
  MSB_HD MSB_INL PList wk_list(int first, int n) {   // `first` = stack index of the word pushed last (low half of lane 0)
    PList l;
    for (int j = 0; j < 3; j++)
      l.w[j] = (unsigned long long)m.sk_ld(first - 2 * j) | ((unsigned long long)m.sk_ld(first - 2 * j - 1) << 32);
    l.set_n(n);
    return l;
  }
  MSB_HD MSB_INL void wk_store_list(int first, const PList& l) {
    for (int j = 0; j < 3; j++) {
      m.sk_st(first - 2 * j, (uint32_t)l.w[j]);
      m.sk_st(first - 2 * j - 1, (uint32_t)(l.w[j] >> 32));
    }
  }
  // ctx_leave(saved) once everything pushed after it has run: the tail of an entity method that switched worlds
  MSB_HD MSB_INL void wk_push_ctx(Wk& k, int sv) {
    if (REM_LISTS && sv >= 0) wk_push(k, mk_hdr(F_CTXLEAVE, 0, sv, 0));
  }

  // ------------------------------------------------------------------------------------------
  // Deferred triggers: Board.push_trigger / pop_trigger (board.py:46-56) and the wrapper that
  // Card.__init_subclass__ puts around every overridden activate_ability (card.py:48-62).
  // ------------------------------------------------------------------------------------------
  MSB_HD MSB_INL void push_trigger(int e, bool src) {
    int n = m.ld8(H_TRIG_N);
    if (n >= TRIG_CAP) {
      set_fault(FAULT_TRIG_STACK);
      return;
    }
    trig_put(n, e, src);
    m.st8(H_TRIG_N, n + 1);
  }
  // entry i of the trigger stack: the slot in the byte, has_source in its top bit -- or, where slot ids need all eight
  // bits (TRIG_WIDE), in bit i of X_TRIGSRC
  MSB_HD MSB_INL void trig_put(int i, int e, bool src) {
    if (TRIG_WIDE) {
      m.st8(OFF_TRIG + i, e);
      uint32_t f = m.ld32(X_TRIGSRC);
      m.st32(X_TRIGSRC, src ? f | (1u << i) : f & ~(1u << i));
    } else
      m.st8(OFF_TRIG + i, e | (src ? 0x80 : 0));
  }
  MSB_HD MSB_INL int trig_slot(int i) const { return m.ld8(OFF_TRIG + i) & TRIG_SLOT; }
  MSB_HD MSB_INL bool trig_src(int i) const { return TRIG_WIDE ? ((m.ld32(X_TRIGSRC) >> i) & 1u) != 0 : (m.ld8(OFF_TRIG + i) & 0x80) != 0; }
  MSB_HD MSB_INL void call_pop_trigger(Wk& k) {
    int n = m.ld8(H_TRIG_N);
    if (n == 0 || m.ld8(H_RESOLVING)) return;
    int e = trig_slot(n - 1);
    bool src = trig_src(n - 1);
    m.st8(H_TRIG_N, n - 1);
    call_run_ability(k, e, -1, PK_NONE, src);
  }
  // wrapped activate_ability (card.py:48-62).  e >= 0: entity slot; e < 0: a spell, `spell` = card | owner << 8.
  // Frame F_RUNAB {hdr: state, e (0xFF = a spell), recursion depth outside | src << 7; w1: spell | pos_pk << 16}:
  // state 0 = the ability has yet to start, 1 = it has returned.  The ability itself starts on run()'s next turn, so
  // that the card code exists once, in the handler.
  MSB_HD MSB_INL void call_run_ability(Wk& k, int e, int spell, int pos_pk, bool src) {
    const int sv = e >= 0 ? ctx_enter(e) : -1;   // the wrapper works on self.player.board (card.py:54-60)
    if (fault()) return;
    wk_push_ctx(k, sv);
    const int d = m.ld8(H_DEPTH);
    if (d >= MAX_DEPTH || !wk_reserve(k)) {
      set_fault(FAULT_DEPTH);
      return;
    }
    m.st8(H_DEPTH, d + 1);
    wk_push(k, (uint32_t)(spell & 0xffff) | ((uint32_t)(pos_pk & 0xff) << 16));
    wk_push(k, mk_hdr(F_RUNAB, 0, e, d | (src ? 0x80 : 0)));
  }
  MSB_HD MSB_INL void h_runab(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1;
    const int d = hdr_b(hdr) & 0x7f;
    if (hdr_st(hdr) == 0) {
      const int e = hdr_a(hdr);
      const bool src = (hdr_b(hdr) & 0x80) != 0;
      const uint32_t w1 = m.sk_ld(top - 1);
      const int pos_pk = (int)((w1 >> 16) & 0xff);
      m.st8(H_RESOLVING, 1);
      if (e != 0xff) {
        M::trace_ability(e_card(e), m.ld8g(eg(e), EO_POS));   // diagnostics hook: nothing in the product
        ability_entity(k, e, pos_pk, src);
      } else {
        const int spell = (int)(w1 & 0xffff);
        M::trace_ability(spell & 0xff, -1);
        ability_spell(k, spell & 0xff, spell >> 8, pos_pk);
      }
      if (fault()) return;
      if (k.sp - 1 != top) {   // still running: its frames are above this one, which resumes behind them
        m.sk_st(top, hdr | (1u << 8));
        return;
      }
    }
    // the ability has returned, card.py:54-60
    m.st8(H_RESOLVING, 0);
    const int n = m.ld8(H_TRIG_N);
    if (n == 0) {
      m.st8(H_DEPTH, d);
      k.sp = top - 1;
      return;
    }
    // the wrapper's trailing pop_trigger() is a tail call in the reference: this frame runs the next deferred ability
    const int e = trig_slot(n - 1);
    const bool src = trig_src(n - 1);
    m.st8(H_TRIG_N, n - 1);
    m.sk_st(top - 1, 0xffffu | ((uint32_t)PK_NONE << 16));
    m.sk_st(top, mk_hdr(F_RUNAB, 0, e, d | (src ? 0x80 : 0)));
  }
  // entity.activate_ability(...) as called by the engine: wrapped iff the class overrides it.
  MSB_HD MSB_INL void call_activate(Wk& k, int e, int pos_pk, bool src) {
    if (e_has_ability(e)) call_run_ability(k, e, -1, pos_pk, src);
  }


  // ------------------------------------------------------------------------------------------
  // Damage / death / statuses
  // ------------------------------------------------------------------------------------------
  // Player.deal_damage / heal, player.py:83-91
  MSB_HD MSB_INL int player_deal_damage(int order, int amount) {
    set_pl_base(order, pl_base(order) - amount);
    return amount;
  }
  MSB_HD MSB_INL void player_heal(int order, int amount) { set_pl_base(order, pl_base(order) + amount); }

  // Unit.deal_damage unit.py:205-219 / Structure.deal_damage structure.py:52-63.  The reference returns the amount
  // dealt; the one caller that uses it (cards/u405.py) takes min(amount, strength) itself (EACH_DMG_HEAL).
  MSB_HD MSB_INL void call_entity_damage(Wk& k, int e, int amount, bool pending, bool src) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    int s = e_str(e);
    if (s - amount < 0) amount = s;
    e_set_dmg(e, amount);
    s -= amount;
    e_set_str(e, s);
    if (!pending && s <= 0) {
      call_destroy(k, e, src);
    } else if (e_trigger(e) == TR_AFTER_SURVIVING && s > 0) {
      push_trigger(e, src);
      if (fault()) return;
      call_pop_trigger(k);
    }
  }
  // X.deal_damage(amount, source=...) where X = board.at(point): unit, structure or Player
  MSB_HD MSB_INL void call_damage(Wk& k, int who, int amount, bool src) {
    if (who >= AT_PLAYER) {
      player_deal_damage(who - AT_PLAYER, amount);
      return;
    }
    if (who < 0) {
      set_fault(FAULT_PY_EXCEPTION);  // None.deal_damage
      return;
    }
    call_entity_damage(k, who, amount, false, src);
  }
  MSB_HD MSB_INL void heal(int who, int amount) {
    if (who >= AT_PLAYER)
      player_heal(who - AT_PLAYER, amount);
    else if (who < 0)
      set_fault(FAULT_PY_EXCEPTION);
    else
      e_set_str(who, e_str(who) + amount);
  }
  // Unit.destroy unit.py:221-231 / Structure.destroy structure.py:65-69.  Frame F_DESTROY_TAIL: what a unit's destroy
  // does once its ON_DEATH ability has returned.
  MSB_HD MSB_INL void call_destroy(Wk& k, int e, bool src) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    if (e_is_unit(e)) {
      board_set(e_pos(e), -1);
      m.st8g(eg(e), EO_PATHN, 0);
      e_set_dmg(e, e_str(e));
      if (e_card_trigger(e) == TR_ON_DEATH) {
        push_trigger(e, src);
        if (fault()) return;
        if (!m.ld8(H_RESOLVING)) {   // pop_trigger() runs an ability now: the rest of destroy waits for it
          wk_push(k, mk_hdr(F_DESTROY_TAIL, 0, 0, 0));
          call_pop_trigger(k);
          return;
        }
      }
      recalc_front_after_destroy();
    } else {
      e_set_dmg(e, e_str(e));
      board_set(e_pos(e), -1);
      recalc_front_after_destroy();
    }
  }
  // status methods, unit.py:239-275.  Calling them on a Structure raises AttributeError.
  MSB_HD MSB_INL bool need_unit(int e) {
    if (e < 0 || e >= AT_PLAYER || !e_is_unit(e)) {
      set_fault(FAULT_PY_EXCEPTION);
      return false;
    }
    return true;
  }
  MSB_HD MSB_INL void freeze(int e) {
    if (need_unit(e)) e_st_add(e, ST_FROZEN);
  }
  MSB_HD MSB_INL void poison(int e) {
    if (!need_unit(e)) return;
    if (e_st(e, ST_VITALIZED) > 0) e_st_remove(e, ST_VITALIZED);
    e_st_add(e, ST_POISONED);
  }
  MSB_HD MSB_INL void vitalize(int e) {
    if (!need_unit(e)) return;
    if (e_st(e, ST_POISONED) > 0) e_st_remove(e, ST_POISONED);
    e_st_add(e, ST_VITALIZED);
  }
  MSB_HD MSB_INL void confuse(int e) {
    if (need_unit(e)) e_st_add(e, ST_CONFUSED);
  }
  MSB_HD MSB_INL void deconfuse(int e) {
    if (need_unit(e)) e_st_remove(e, ST_CONFUSED);
  }
  MSB_HD MSB_INL void disable(int e) {  // unit.py:269-271: only classes that override the ability
    if (need_unit(e) && e_has_ability(e)) e_st_add(e, ST_DISABLED);
  }

  // ------------------------------------------------------------------------------------------
  // Movement
  // ------------------------------------------------------------------------------------------
  // Unit.set_path, unit.py:78-122
  MSB_HD MSB_INL void set_path(int e, bool on_play) {
    const int sv = ctx_enter(e);
    set_path_impl(e, on_play);
    ctx_leave(sv);
  }
  MSB_HD MSB_A_SETPATH void set_path_impl(int e, bool on_play) {
    MSB_SCOPE(PS_SET_PATH);
    P position = e_pos(e);
    int confused_cached = e_st(e, ST_CONFUSED);
    int owner = e_owner(e);
    bool is_local = owner == local();
    int steps = on_play ? e_mov(e) : 1;
    uint32_t packed = 0;
    int n = 0;
    unsigned long long dests = 0;   // up to 8 packed destinations, one byte each (kept in registers)
    int nd = 0;
    for (int k = 0; k < steps; k++) {
      P dest{position.x, position.y + (is_local ? -1 : 1)};
      int nxt = at(dest);
      if (confused_cached > 0) {
        int dx;
        if (position.x == 0)
          dx = 1;                                  // choice([1]): no draw
        else if (position.x == 3)
          dx = -1;                                 // choice([-1]): no draw
        else
          dx = rng_randint(0, 2) ? 1 : -1;         // choice([-1, 1])
        dest = P{position.x + dx, position.y};
        confused_cached--;
      } else if (on_play && !e_ff(e) && dest.y != (is_local ? -1 : 5) &&
                 (nxt == AT_NONE || (nxt < AT_PLAYER && e_owner(nxt) == owner))) {
        P lp{position.x - 1, position.y}, rp{position.x + 1, position.y};
        int left = position.x > 0 ? at(lp) : AT_NONE;
        int right = position.x < 3 ? at(rp) : AT_NONE;
        bool left_ok = left >= 0 && left < AT_PLAYER && e_owner(left) != owner;
        bool right_ok = right >= 0 && right < AT_PLAYER && e_owner(right) != owner;
        int lpk = p_pack(lp), rpk = p_pack(rp);
        bool lseen = false, rseen = false;
        for (int i = 0; i < nd; i++) {
          int dv = (int)((dests >> (8 * i)) & 0xff);
          if (dv == lpk) lseen = true;
          if (dv == rpk) rseen = true;
        }
        if (position.x <= 1) {
          if (right_ok && !rseen)
            dest = rp;
          else if (left_ok && !lseen)
            dest = lp;
        } else {
          if (left_ok && !lseen)
            dest = lp;
          else if (right_ok && !rseen)
            dest = rp;
        }
      }
      // Destinations past the first off-board one are never read (move returns at the base hit);
      // they are clamped so that they stay encodable.
      P enc = dest;
      if (enc.y < -1) enc.y = -1;
      if (enc.y > 5) enc.y = 5;
      if (nd < 8) dests |= (unsigned long long)p_pack(enc) << (8 * nd++);
      if (n < PATH_CAP) {
        packed |= (uint32_t)p_pack(enc) << (8 * n);
        n++;
      } else {
        set_fault(FAULT_CAP_PATH);
      }
      position = dest;
    }
    e_set_path(e, packed);
    m.st8g(eg(e), EO_PATHN, n);
  }

  // Unit.move, unit.py:124-203.  Frame F_MOVE {hdr: state, e, recursion depth outside | MV_PLAYED; w1: the path list
  // bound when the loop started (fact #5); w2: i | n << 3 | move_id << 8 | target << 16 | flags << 24 (1 is_attacked,
  // 2 target_pending, 4 local_pending); w3: the target's strength before the fight}.  The states are the places where
  // the reference's move() is waiting for a nested call; MV_START is the call itself (a move pushed below the ability
  // that runs first: Unit.play).  MV_PLAYED: this is the move at the end of Unit.play (unit.py:74-76), which clears
  // resolving_play behind it.
  enum : int { MV_ENTRY = 0, MV_POISONED, MV_BEFORE_MOVING, MV_STEP, MV_BASE_HIT, MV_FIGHT, MV_STRUCK, MV_STRUCK_BACK,
               MV_TARGET_DEAD, MV_SELF_DEAD, MV_AFTER_ATTACK, MV_DONE, MV_START };
  static constexpr int MV_PLAYED = 0x80;
  // the entry of move(): the world of the unit, the recursion guard.  Returns the depth outside, -1 if the step has faulted.
  MSB_HD MSB_INL int move_enter(Wk& k, int e) {
    const int sv = ctx_enter(e);
    if (fault()) return -1;
    wk_push_ctx(k, sv);
    const int d = m.ld8(H_DEPTH);
    if (d >= MAX_DEPTH || !wk_reserve(k)) {
      set_fault(FAULT_DEPTH);
      return -1;
    }
    m.st8(H_DEPTH, d + 1);
    return d;
  }
  // (a move's three data words hold nothing until it first waits for a nested call: only the header is written here)
  MSB_HD MSB_INL void call_move(Wk& k, int e, int played = 0) {
    const int d = move_enter(k, e);
    if (d < 0) return;
    k.sp += 3;
    wk_push(k, mk_hdr(F_MOVE, MV_ENTRY, e, d | played));
  }
  // a move that starts when the frames pushed after it have run
  MSB_HD MSB_INL void call_move_later(Wk& k, int e, int played) {
    k.sp += 3;
    wk_push(k, mk_hdr(F_MOVE, MV_START, e, played));
  }
  MSB_HD MSB_INL void h_move(Wk& k, uint32_t hdr) {
    if (hdr_st(hdr) == MV_START) {
      // the frame holds nothing yet: take it off, enter (which may leave a world mark where it was), put it back
      const int e0 = hdr_a(hdr), played = hdr_b(hdr) & MV_PLAYED;
      k.sp -= 4;
      const int d0 = move_enter(k, e0);
      if (d0 < 0) return;
      k.sp += 4;
      hdr = mk_hdr(F_MOVE, MV_ENTRY, e0, d0 | played);
    }
    const int top = k.sp - 1;
    const int e = hdr_a(hdr), d = hdr_b(hdr) & 0x7f;
    const int played_tail = hdr_b(hdr) & MV_PLAYED;
    uint32_t path = 0, w2 = 0;
    int cached = 0;
    if (hdr_st(hdr) != MV_ENTRY) {   // a move that has waited: its locals
      path = m.sk_ld(top - 1);
      w2 = m.sk_ld(top - 2);
      cached = (int)(int16_t)(m.sk_ld(top - 3) & 0xffffu);
    }
    int i = (int)(w2 & 7), n = (int)((w2 >> 3) & 7), current_id = (int)((w2 >> 8) & 0xff), target = (int)((w2 >> 16) & 0xff);
    int flags = (int)(w2 >> 24);
    const int trig = e_trigger(e);
    int next = MV_DONE;
    P dest{0, 0};
    int owner = 0, tp = 0;
#define MV_WAIT(s_)  \
  do {               \
    next = (s_);     \
    goto wait;       \
  } while (0)
#define MV_CALLED(s_)                        \
  do {                                       \
    if (fault()) return;                     \
    if (k.sp - 1 != top) MV_WAIT(s_);        \
  } while (0)
    switch (hdr_st(hdr)) {
      case MV_ENTRY:
        current_id = (m.ld8g(eg(e), EO_MOVEID) + 1) & 0xff;
        m.st8g(eg(e), EO_MOVEID, current_id);
        if (phase() == PH_TURN_START) {
          if (e_st(e, ST_POISONED) > 0) {
            call_entity_damage(k, e, 1, false, false);
            MV_CALLED(MV_POISONED);
          } else if (e_st(e, ST_VITALIZED) > 0)
            e_set_str(e, e_str(e) + 1);
        }
        // fall through
      case MV_POISONED:
        if (phase() == PH_TURN_START && e_frozen(e)) {
          e_st_remove(e, ST_FROZEN);
          goto done;
        }
        if (m.ld8g(eg(e), EO_PATHN) == 0) goto done;
        if (trig == TR_BEFORE_MOVING && !e_disabled(e)) {
          call_run_ability(k, e, -1, PK_NONE, true);
          if (fault()) return;
          MV_WAIT(MV_BEFORE_MOVING);
        }
        // fall through
      case MV_BEFORE_MOVING:
        if (e_frozen(e)) goto done;
        // `for destination in self.path` iterates the list object bound now (fact #5)
        n = m.ld8g(eg(e), EO_PATHN);
        path = e_path(e);
        i = 0;
        // fall through
      case MV_STEP:
        if (i >= n) goto done;
        dest = p_unpack((path >> (8 * i)) & 0xff);
        owner = e_owner(e);  // self.player is re-read by the reference; convert() may change it
        flags = 0;
        if (dest.y < 0 || dest.y > 4) {
          if (trig == TR_BEFORE_ATTACKING && !e_disabled(e)) {
            call_run_ability(k, e, -1, p_pack(dest), true);
            if (fault()) return;
            MV_WAIT(MV_BASE_HIT);
          }
          goto base_hit;
        }
        target = at(dest);
        if (target != AT_NONE && e_owner(target) == owner && dest.x == e_pos(e).x) goto done;
        if (!(target != AT_NONE && (e_confused(e) || e_owner(target) != owner))) goto advance;
        if (trig == TR_BEFORE_ATTACKING && !e_disabled(e)) {
          call_run_ability(k, e, -1, p_pack(dest), true);
          if (fault()) return;
          MV_WAIT(MV_FIGHT);
        }
        goto fight;
      case MV_BASE_HIT:
        dest = p_unpack((path >> (8 * i)) & 0xff);
      base_hit:
        MSB_COUNT_EVENT(50);
        tp = dest.y < 0 ? remote() : local();
        player_deal_damage(tp, e_str(e));
        if (pl_base(tp) > 0) {
          call_destroy(k, e, false);
          MV_CALLED(MV_DONE);
        }
        goto done;
      case MV_FIGHT:
        dest = p_unpack((path >> (8 * i)) & 0xff);
      fight:
        MSB_COUNT_EVENT(51);
        target = at(dest);
        if (target == AT_NONE) goto advance;
        cached = e_str(target);   // target_strength_cached
        flags = ((e_trigger(target) == TR_ON_DEATH && !e_disabled(target)) ? 2 : 0) | ((trig == TR_ON_DEATH && !e_disabled(e)) ? 4 : 0);
        call_entity_damage(k, target, e_str(e), (flags & 2) != 0, false);
        MV_CALLED(MV_STRUCK);
        // fall through
      case MV_STRUCK:
        call_entity_damage(k, e, cached, (flags & 4) != 0, false);
        MV_CALLED(MV_STRUCK_BACK);
        // fall through
      case MV_STRUCK_BACK:
        if (e_str(target) <= 0 && (flags & 2)) {
          call_destroy(k, target, false);
          MV_CALLED(MV_TARGET_DEAD);
        }
        // fall through
      case MV_TARGET_DEAD:
        if (e_str(e) <= 0 && (flags & 4)) {
          call_destroy(k, e, false);
          MV_CALLED(MV_SELF_DEAD);
        }
        // fall through
      case MV_SELF_DEAD:
        flags |= 1;   // is_attacked
      advance:
        MSB_COUNT_EVENT(52);
        dest = p_unpack((path >> (8 * i)) & 0xff);
        if (current_id != m.ld8g(eg(e), EO_MOVEID)) goto done;
        if (at(dest) == AT_NONE && e_str(e) > 0) {
          board_set(e_pos(e), -1);
          board_set(dest, e);
          const int o = e_owner(e);
          if (pl_front(o) > dest.y) set_pl_front(o, dest.y > 1 ? dest.y : 1);
          if ((flags & 1) && trig == TR_AFTER_ATTACKING && !e_disabled(e)) {
            call_run_ability(k, e, -1, PK_NONE, true);
            if (fault()) return;
            MV_WAIT(MV_AFTER_ATTACK);
          }
          if (e_confused(e)) e_st_remove(e, ST_CONFUSED);
        }
        goto next_step;
      case MV_AFTER_ATTACK:
        if (e_confused(e)) e_st_remove(e, ST_CONFUSED);
      next_step:
        i++;
        if (i < n) MV_WAIT(MV_STEP);   // the next destination: a backward jump, through run()
        goto done;
      default:   // MV_DONE
        break;
    }
  done:
    m.st8(H_DEPTH, d);
    if (played_tail) e_set_flag(e, EF_RESOLVING_PLAY, false);   // Unit.play's last line
    k.sp = top - 3;
    return;
  wait:
    m.sk_st(top - 1, path);
    m.sk_st(top - 2, (uint32_t)(i & 7) | ((uint32_t)(n & 7) << 3) | ((uint32_t)(current_id & 0xff) << 8) | ((uint32_t)(target & 0xff) << 16) |
                         ((uint32_t)(flags & 0xff) << 24));
    m.sk_st(top - 3, (uint32_t)(cached & 0xffff));
    m.sk_st(top, mk_hdr(F_MOVE, next, e, d | played_tail));
#undef MV_WAIT
#undef MV_CALLED
  }

  // Unit.play, unit.py:66-76: the ON_PLAY ability, then the move, whose frame also carries play()'s last line (MV_PLAYED)
  MSB_HD MSB_INL void call_unit_play(Wk& k, int e, P position) {
    e_set_flag(e, EF_RESOLVING_PLAY, true);
    board_set(position, e);
    set_path(e, true);
    if (fault()) return;
    if (e_card_trigger(e) == TR_ON_PLAY) {
      call_move_later(k, e, MV_PLAYED);   // below the ability: starts when the ability has returned
      call_run_ability(k, e, -1, PK_NONE, true);
    } else {
      call_move(k, e, MV_PLAYED);
    }
  }
  // Structure.play, structure.py:45-50
  MSB_HD MSB_INL void call_structure_play(Wk& k, int e, P position) {
    board_set(position, e);
    if (e_card_trigger(e) == TR_ON_PLAY) call_run_ability(k, e, -1, PK_NONE, true);
  }
  MSB_HD MSB_INL bool e_resolving_play(int e) const { return (e_flags(e) & EF_RESOLVING_PLAY) != 0; }
  // Unit.gain_speed, unit.py:277-280
  MSB_HD MSB_INL void gain_speed(int e, int amount) {
    int mv = e_mov(e);
    m.st8g(eg(e), EO_MOV, mv + amount);
    set_path(e, e_resolving_play(e));
    m.st8g(eg(e), EO_MOV, mv);
  }
  // Unit.command, unit.py:282-289.  Frame F_CMD_TAIL {hdr: e, fixedly_forward as it was}: behind the move.
  MSB_HD MSB_INL void call_command(Wk& k, int e) {
    if (!need_unit(e)) return;
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    const bool ff = e_ff(e);
    e_set_flag(e, EF_FF, true);
    set_path(e, false);
    if (fault()) return;
    wk_push(k, mk_hdr(F_CMD_TAIL, 0, e, ff ? 1 : 0));
    call_move(k, e);
  }
  // Unit.convert, unit.py:291-293
  MSB_HD MSB_INL void convert(int e) {
    int o = e_owner(e);
    int no = opponent_of(o);
    e_set_flag(e, EF_OWNER, no != 0);
    set_path(e, e_resolving_play(e));
  }
  // Unit.teleport, unit.py:373-382
  MSB_HD MSB_INL void teleport(int e, P dest) {
    const int sv = ctx_enter(e);
    teleport_impl(e, dest);
    ctx_leave(sv);
  }
  MSB_HD MSB_A_MISC void teleport_impl(int e, P dest) {
    if (at(dest) == AT_NONE) {
      board_set(e_pos(e), -1);
      board_set(dest, e);
      int o = e_owner(e);
      if (pl_front(o) > dest.y) set_pl_front(o, dest.y > 1 ? dest.y : 1);
      set_path(e, e_resolving_play(e));
    }
  }
  // Unit.push (away from `from`) unit.py:318-339 and Unit.pull (towards) unit.py:295-316
  MSB_HD MSB_INL void push_pull(int e, P from, bool is_push) {
    if (!need_unit(e)) return;
    const int sv = ctx_enter(e);
    push_pull_impl(e, from, is_push);
    ctx_leave(sv);
  }
  MSB_HD MSB_A_MISC void push_pull_impl(int e, P from, bool is_push) {
    P pos = e_pos(e);
    int dx = 0, dy = 0;
    if (from.y < pos.y)
      dy = is_push ? 1 : -1;
    else if (from.y > pos.y)
      dy = is_push ? -1 : 1;
    else if (from.x < pos.x)
      dx = is_push ? 1 : -1;
    else if (from.x > pos.x)
      dx = is_push ? -1 : 1;
    if (dx != 0 || dy != 0) {
      for (;;) {
        P nx{e_pos(e).x + dx, e_pos(e).y + dy};
        if (!p_valid(nx)) break;
        if (at(nx) != AT_NONE) return;  // NB: returns before the front-line update (unit.py:331-332)
        board_set(e_pos(e), -1);
        board_set(nx, e);
      }
    }
    int o = e_owner(e);
    int y = e_pos(e).y;
    if (pl_front(o) > y) set_pl_front(o, y > 1 ? y : 1);
  }
  // Unit.force_attack, unit.py:341-371 (the move at its end is a tail call)
  MSB_HD MSB_INL void call_force_attack(Wk& k, int e, P dest) {
    const int sv = ctx_enter(e);
    if (fault()) return;
    wk_push_ctx(k, sv);
    P pos = e_pos(e);
    if ((dest.x != pos.x && dest.y != pos.y) || at(dest) == AT_NONE) return;
    bool vertical = dest.x == pos.x;
    int fixed = vertical ? pos.x : pos.y;
    int start = vertical ? pos.y : pos.x;
    int end = vertical ? dest.y : dest.x;
    int delta = end > start ? 1 : -1;
    uint32_t packed = 0;
    int n = 0;
    for (int i = start + delta; i != end + delta; i += delta) {
      P pt = vertical ? P{fixed, i} : P{i, fixed};
      if (i != end && at(pt) != AT_NONE) return;
      if (n < PATH_CAP) packed |= (uint32_t)p_pack(pt) << (8 * n);
      n++;
    }
    if (n > PATH_CAP) {
      set_fault(FAULT_CAP_PATH);
      return;
    }
    if (n > 0) {
      e_set_path(e, packed);
      m.st8g(eg(e), EO_PATHN, n);
      call_move(k, e);
    }
  }
  // Board.spawn_token_unit, board.py:298-311 (types always given by the cards)
  MSB_HD MSB_A_MISC int spawn_token_unit(int owner, P position, int strength, int unit_type) {
    int e = new_entity(TOKEN_UNIT_BASE + unit_type, owner, strength, 1, false);
    if (fault()) return e;
    board_set(position, e);
    calculate_front_line(owner);
    return e;
  }
  // Unit.respawn unit.py:384-402 / Structure.respawn structure.py:77-89: a fresh object of the same
  // class with the given strength is written onto the tile (whatever was there is overwritten).
  MSB_HD MSB_INL void respawn(int e, P position, int strength) {
    const int sv = ctx_enter(e);
    respawn_impl(e, position, strength);
    ctx_leave(sv);
  }
  MSB_HD MSB_A_MISC void respawn_impl(int e, P position, int strength) {
    int c = e_card(e);
    int ne;
    if (c < NUM_CARDS)
      ne = new_entity(c, e_owner(e), strength, g_cards[c].movement, g_cards[c].ff != 0);
    else
      ne = new_entity(c, e_owner(e), strength, e_mov(e), e_ff(e));
    if (fault()) return;
    board_set(position, ne);
  }

  // ------------------------------------------------------------------------------------------
  // Player: hand / deck (player.py:46-81)
  // ------------------------------------------------------------------------------------------
  // Card equality used by list.remove: Unit/Structure compare (card_id, player, position)
  // (unit.py:25-26, structure.py:18-19); Spells compare identity (card.py:22-23).
  MSB_HD MSB_INL bool card_eq_by_id(int card) const { return card >= NUM_CARDS || g_cards[card].kind != KIND_SPELL; }

#if defined(MSB_EXT) && MSB_EXT
  // .position of a card object that has been on the board (b305 returns the structure OBJECT to the hand): the entity's
  // recorded position while it is there, the one it had when it left afterwards
  MSB_HD MSB_INL int inst_position(int o, int ref) const {
    int fl = m.ld8(ref + 2), x = m.ld8(ref + 3);
    if (fl & CF_ALIAS) return m.ld8g(eg(x), EO_POS);
    return m.ld8(pl(o, P_IPOS + (ref - pl(o, P_INST)) / 4));
  }
#endif
  // list.remove(target): index of the first element EQUAL to the one at `idx`.  Unit/Structure
  // equality is (card_id, player, position); instances that came back from the board (b305) have a
  // position, fresh cards have None: comparing None with a Point raises (point.py:6-7).
  MSB_HD MSB_INL int first_equal(int o, bool in_hand, int idx) {
    int tref = in_hand ? hand_ref(o, idx) : deck_ref(o, idx);
    int card = m.ld8(tref);
    int positioned = m.ld8(tref + 2) & (CF_ALIAS | CF_STR);
    for (int i = 0; i < idx; i++) {
      int r = in_hand ? hand_ref(o, i) : deck_ref(o, i);
      if (r == tref) return i;   // the very same object listed earlier (extended record only)
      if (!card_eq_by_id(card) || m.ld8(r) != card) continue;
      int pi = m.ld8(r + 2) & (CF_ALIAS | CF_STR);
      if (!pi && !positioned) return i;
#if defined(MSB_EXT) && MSB_EXT
      if (pi && positioned) {   // two objects that have both been on the board: equal iff their positions are
        int pa = inst_position(o, r), pb = inst_position(o, tref);
        if (pa != IPOS_UNKNOWN && pb != IPOS_UNKNOWN) {
          if (pa == pb) return i;
          continue;
        }
      }
#endif
      set_fault((pi && positioned) ? FAULT_UNSUPPORTED : FAULT_PY_EXCEPTION);
      return idx;
    }
    return idx;
  }

  // Player.draw, player.py:46-52: numpy choice(deck, size=1, p=w/sum(w))
  MSB_HD MSB_A_DRAW void draw(int o, int amount) {
    MSB_SCOPE(PS_DRAW);
    for (int k = 0; k < amount; k++) {
      int n = pl_deck_n(o);
      if (n == 0) {
        set_fault(FAULT_PY_EXCEPTION);  // choice over an empty deck raises
        return;
      }
      double sum = 0.0;  // Python sum(): 0 + w0 + w1 ... left to right
      for (int i = 0; i < n; i++) sum = sum + deck_w(o, i);
      // cdf = cumsum(w/sum); cdf /= cdf[-1]; idx = searchsorted(cdf, u, 'right').  The running sum is
      // recomputed in the second pass (same operations, same order) instead of being kept in an array.
      double last = 0.0;
      for (int i = 0; i < n; i++) {
        double p = deck_w(o, i) / sum;
        last = (i == 0) ? p : last + p;   // ndarray.cumsum: sequential
      }
      double u = rng_random_sample();
      int idx = 0;
      double acc = 0.0;
      for (; idx < n; idx++) {
        double p = deck_w(o, idx) / sum;
        acc = (idx == 0) ? p : acc + p;
        if (!(acc / last <= u)) break;
      }
      if (idx >= n) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
      set_deck_age(o, idx, 0);             // choice.weight = 1
      hand_push_from_deck(o, idx);         // self.hand.append(choice)
      if (fault()) return;
      int j = first_equal(o, false, idx);  // self.deck.remove(choice): first EQUAL element
      if (fault()) return;
      deck_remove_at(o, j);
    }
  }
  // Player.fill_hand, player.py:54-55
  MSB_HD MSB_INL void fill_hand(int o) {
    int need = 4 - pl_hand_n(o);
    if (need > 0) draw(o, need);
  }
  // Player.discard, player.py:57-66 (reweight: w*1.6+100 for every deck card)
  MSB_HD MSB_INL void discard(int o, int hand_index) {
    reweight(o);
    if (fault()) return;
    uint32_t target = hand_handle(o, hand_index);
    int fl = hand_flags(o, hand_index);
    int j = first_equal(o, true, hand_index);   // hand.remove(target): first equal
    if (fault()) return;
    hand_remove_at(o, j);
    if (!(fl & CF_SINGLE_USE)) deck_push_handle(o, target);
  }
  // Board.add_to_history, board.py:324-325 (only the last four are observable)
  MSB_HD MSB_INL void add_history(int owner, int card) {
    // four {owner, card} byte pairs, oldest first, as one 64-bit word: append, or drop the oldest and append
    int n = m.ld8(H_HIST_N);
    uint64_t h = m.ld64(H_HIST);
    const uint64_t entry = (uint64_t)(owner & 0xff) | ((uint64_t)(card & 0xff) << 8);
    if (n < 4) {
      h = (h & ~(0xffffull << (16 * n))) | (entry << (16 * n));
      m.st8(H_HIST_N, n + 1);
    } else {
      h = (h >> 16) | (entry << 48);
    }
    m.st64(H_HIST, h);
  }
  // Player.play, player.py:68-77.  has_pos=false <=> position None
  MSB_HD MSB_INL void call_player_play(Wk& k, int o, int index, P position, bool has_pos) {
    MSB_SCOPE(PS_PLAYER_PLAY);
    int card = hand_card(o, index), fl = hand_flags(o, index);
    int strength = inst_strength(card, fl, hand_x(o, index));   // target.copy() copies the instance's strength
    add_history(o, card);
    discard(o, index);
    if (fault()) return;
    const CardInfo& ci = g_cards[card];
    if (ci.kind == KIND_UNIT) {
      int e = new_entity(card, o, strength, ci.movement, (fl & CF_FF) != 0);   // target.copy()
      if (fault()) return;
      if (!has_pos) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
      call_unit_play(k, e, position);
    } else if (ci.kind == KIND_STRUCT) {
      int e = new_entity(card, o, strength, 0, false);
      if (fault()) return;
      if (fl & CF_SINGLE_USE) e_set_flag(e, EF_SINGLE_USE, true);
      if (!has_pos) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
      call_structure_play(k, e, position);
    } else {
      call_spell_play(k, card, o, position, has_pos);
    }
  }
  // Spell.play, spell.py:22-24
  MSB_HD MSB_INL void call_spell_play(Wk& k, int card, int o, P position, bool has_pos) {
    const CardInfo& ci = g_cards[card];
    bool go = true;
    if (ci.tgt.has) {
      Tgt t = mk_tgt(ci.tgt);
      PList l = get_targets(cp(), t, PK_NONE);
      go = has_pos && l.has(position);
      // `None in [Point...]` evaluates Point.__eq__(None) -> AttributeError when the list is non-empty
      if (!has_pos && l.n() > 0) {
        set_fault(FAULT_PY_EXCEPTION);
        return;
      }
    }
    if (go) call_run_ability(k, -1, card | (o << 8), has_pos ? p_pack(position) : PK_NONE, true);
  }
  // Player.cycle, player.py:79-81
  MSB_HD MSB_INL void cycle(int o, int hand_index) {
    discard(o, hand_index);
    if (fault()) return;
    draw(o, 1);
  }

  // ------------------------------------------------------------------------------------------
  // Turn structure
  // ------------------------------------------------------------------------------------------
  // Board.flip, board.py:94-115.  H_TOPLAY has already been toggled by the caller, which swaps
  // local/remote; entity.player keeps the same PlayerOrder through the "ownership swap".
  MSB_HD MSB_A_TURN void flip() {
    set_pl_front(0, 4 - pl_front(0));
    set_pl_front(1, 4 - pl_front(1));
    // 180-degree rotation: new row y = byte-reversed old row 4-y
    const uint32_t r0 = board_row(0), r1 = board_row(1), r2 = board_row(2), r3 = board_row(3), r4 = board_row(4);
    for (int y = 0; y < 5; y++) {
      uint32_t v = y == 0 ? r4 : y == 1 ? r3 : y == 2 ? r2 : y == 3 ? r1 : r0;
      v = (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
      m.st32(OFF_BOARD + 4 * y, v);
      for (int x = 0; x < 4; x++) {
        uint32_t s = (v >> (8 * x)) & 0xff;
        if (s != (uint32_t)SLOT_NONE) {
          m.st8g(eg((int)s), EO_POS, y * 4 + x);
          // board.py:108-115: every on-board entity's .player is re-bound to the board's own players -- an entity that
          // still belonged to a frozen world (a restored nested b005 memory) joins the real game here
          if (REM_LISTS) m.st8(E_HOME + (int)s, 0);
        }
      }
    }
  }
  // Board.to_next_turn, board.py:117-145.  Frame F_TURN {hdr: state, i, ns; six words: the snapshot of entity OBJECTS
  // being iterated (fact #6), as slot ids}: state 0 = the friendly structures' TURN_START abilities, 1 = the units' moves.
  MSB_HD MSB_INL void call_next_turn(Wk& k) {
    m.st8(H_PHASE, PH_TURN_END);
    int ender = cp();
    fill_hand(ender);
    if (fault()) return;
    // No card in the reference has a TURN_END trigger (structure.py:8 default is [TURN_START],
    // b305 is [ON_PLAY]); the TURN_END loop (board.py:121-123) never fires an ability.
    calculate_front_line(local());
    calculate_front_line(remote());
    m.st16(pl(ender, P_MAXMANA), pl_maxmana(ender) + 1);
    set_pl_mana(0, pl_maxmana(0));
    set_pl_mana(1, pl_maxmana(1));
    m.st8(H_PHASE, PH_TURN_START);
    int ncp = (ender == local()) ? remote() : local();
    m.st8(H_CP, ncp);
    m.st8(pl(ncp, P_FLAGS), m.ld8(pl(ncp, P_FLAGS)) | 3);
    PList hs;
    hs.clear();
    PList snap = get_targets(ncp, mk_tgt(TK_STRUCTURE, TS_FRIENDLY), PK_NONE);
    int ns = snap.n();
    for (int i = 0; i < ns; i++) hs.set8(i, at(snap.at(i)));
    wk_push_list(k, hs);
    wk_push(k, mk_hdr(F_TURN, 0, 0, ns));
  }
  MSB_HD MSB_INL void h_turn(Wk& k, const uint32_t hdr) {
    const int top = k.sp - 1;
    int i = hdr_a(hdr), ns = hdr_b(hdr);
    PList hs = wk_list(top - 1, 0);
    if (hdr_st(hdr) == 0) {
      while (i < ns) {
        const int s = hs.get8(i);
        i++;
        // structure.is_at_turn_start: token structures and b001 run the empty base ability
        if (e_card_trigger(s) == TR_TURN_START) {
          m.sk_st(top, mk_hdr(F_TURN, 0, i, ns));
          call_run_ability(k, s, -1, m.ld8g(eg(s), EO_POS) /*unused*/, true);
          return;
        }
      }
      // the units as they stand once every structure has acted
      PList snap = get_targets(cp(), mk_tgt(TK_UNIT, TS_FRIENDLY), PK_NONE);
      ns = snap.n();
      for (int j = 0; j < ns; j++) hs.set8(j, at(snap.at(j)));
      wk_store_list(top - 1, hs);
      i = 0;
    }
    if (i < ns) {
      const int u = hs.get8(i);
      m.sk_st(top, mk_hdr(F_TURN, 1, i + 1, ns));
      set_path(u, false);
      if (fault()) return;
      call_move(k, u);
      return;
    }
    m.st8(H_PHASE, PH_PLAY);
    k.sp = top - 6;
  }


  // Stormbound.have_winner, games/stormbound.py:560-561 (strict: a base at exactly 0 is alive)
  MSB_HD MSB_INL bool have_winner() const { return pl_base(0) < 0 || pl_base(1) < 0; }

  // Stormbound.legal_actions + Action.to_int, games/stormbound.py:528-557, 258-290 -> 156-bit mask
  // (three 64-bit words in a register vector)
  MSB_HD MSB_A_LEGAL msb_u64x4 legal_mask_v() {
    unsigned long long m0 = 0, m1 = 0, m2 = 0;
    int lo = local();
    int hn = pl_hand_n(lo), mana = pl_mana(lo), fl = pl_front(lo);
    bool any_play = false;
#define MSB_SETBIT(a)                                   \
  do {                                                  \
    int a_ = (a);                                       \
    if (a_ < 64) m0 |= 1ull << a_;                      \
    else if (a_ < 128) m1 |= 1ull << (a_ - 64);         \
    else if (a_ < 156) m2 |= 1ull << (a_ - 128);        \
  } while (0)
    // empty tiles within the front line, as the 16 PLACE offsets (4-y)*4+x of one card (y = 4..max(fl,1));
    // tiles of row 0 are not enumerated by Action.to_int and stay 155 (games/stormbound.py:262-271)
    const uint32_t empty = empty_mask();
    uint32_t place = 0;
    for (int y = 4; y >= 1; y--)
      if (y >= fl) place |= ((empty >> (4 * y)) & 0xFu) << ((4 - y) * 4);
    const bool place_row0 = fl <= 0 && (empty & 0xFu) != 0;
    for (int c = 0; c < hn; c++) {
      const uint32_t inst = m.ld32(hand_ref(lo, c));   // {card, cost, flags, x}
      if ((int)((inst >> 8) & 0xff) > mana) continue;
      int card = (int)(inst & 0xff);
      const CardInfo& ci = g_cards[card];
      if (ci.kind != KIND_SPELL) {
        if (place) {
          if (c < 4) m0 |= (unsigned long long)place << (16 * c);
          else m1 |= (unsigned long long)place << (16 * c - 64);
          any_play = true;
        }
        if (place_row0) {
          MSB_SETBIT(155);
          any_play = true;
        }
      } else if (!ci.tgt.has) {
        MSB_SETBIT(64 + 21 * c);
        any_play = true;
      } else {
        PList l = get_targets(cp(), mk_tgt(ci.tgt), PK_NONE);
        int ln = l.n();
        for (int i = 0; i < ln; i++) {
          P p = l.at(i);
          MSB_SETBIT(p_valid(p) ? 65 + 21 * c + (4 - p.y) * 4 + p.x : 155);
          any_play = true;
        }
      }
    }
    if (m.ld8(pl(lo, P_FLAGS)) & 1)
      for (int c = 0; c < hn; c++) MSB_SETBIT(148 + c);
    if (!any_play) MSB_SETBIT(155);
#undef MSB_SETBIT
    return msb_u64x4{m0, m1, m2, 0ull};
  }
  MSB_HD MSB_INL void legal_mask(uint64_t mask[3]) {
    msb_u64x4 r = legal_mask_v();
    mask[0] = r[0];
    mask[1] = r[1];
    mask[2] = r[2];
  }

  // Stormbound.step, games/stormbound.py:318-373 (without the observation; see observe.inc).
  // The caller guarantees `action` is in legal_actions().  Returns reward | done << 1 as the reference
  // computes them.  A play leaves frames for run(); so does passing the turn on, which has nothing before it.
#if defined(MSB_STUDY_REPEAT)
  // study build (scripts/step_cost.sh): a step that stops early -- cut 1 after begin_step, 2 before run(), 3.. after
  // cut - 2 rounds of run()'s loop; its result is thrown away by the caller
  MSB_HD MSB_INL int step(int action, int cut = 0) { return step_impl(action, cut); }
  MSB_HD MSB_A_STEP int step_impl(int action, int cut) {
#define MSB_STUDY_CUT(n_) if (cut == (n_)) return 0;
#define MSB_STUDY_LIMIT (cut >= 3 ? cut - 2 : 1 << 30)
#else
  MSB_HD MSB_INL int step(int action) { return step_impl(action); }
  MSB_HD MSB_A_STEP int step_impl(int action) {
#define MSB_STUDY_CUT(n_)
#endif
    MSB_SCOPE(PS_STEP);
    Wk k{0, 0, 0, 0};
    begin_step();
    if (fault()) return 0;
    MSB_STUDY_CUT(1)
    int lo = local();
    if (action < 148) {
      // PLACE: card = a//16, tile = a%16 over y=4..1,x=0..3.  USE: card = (a-64)//21, idx = (a-64)%21; the
      // countdown executes at the idx-th tile of y=4..0,x=0..3 -- one tile after the one Action.to_int
      // encoded (fact #2); idx==20 falls off the loop: nothing happens at all.
      bool place = action < 64;
      int ci = place ? action >> 4 : (action - 64) / 21;
      int idx = place ? action & 15 : (action - 64) % 21;
      if (idx < 20) {
        P pos{idx & 3, 4 - (idx >> 2)};
        bool has_pos = place || g_cards[hand_card(lo, ci)].tgt.has != 0;
        set_pl_mana(lo, pl_mana(lo) - hand_cost(lo, ci));
        call_player_play(k, lo, ci, pos, has_pos);
      }
    } else if (action < 152) {
      cycle(lo, action - 148);
      m.st8(pl(lo, P_FLAGS), m.ld8(pl(lo, P_FLAGS)) & ~1);
    } else if (action < 155) {
      int ci = action - 151;
#if defined(MSB_EXT) && MSB_EXT
      int a = hand_id(lo, ci), b = hand_id(lo, 0);
      m.st8(pl(lo, P_HAND + ci), b);
      m.st8(pl(lo, P_HAND), a);
#else
      uint32_t a = m.ld32(pl(lo, P_HAND + 4 * ci)), b = m.ld32(pl(lo, P_HAND));
      m.st32(pl(lo, P_HAND + 4 * ci), b);
      m.st32(pl(lo, P_HAND), a);
#endif
      m.st8(pl(lo, P_FLAGS), m.ld8(pl(lo, P_FLAGS)) & ~2);
    } else if (action == 155) {
      // done = have_winner() or len(legal_actions()) == 0; legal_actions() is never empty (PASS)
      k.result = (pl_base(remote()) <= 0 ? 1 : 0) | (have_winner() ? 2 : 0);
      m.st8(H_TOPLAY, lo ^ 1);
      flip();
      k.sp = next_turn_out(k.sp);
    }
    MSB_STUDY_CUT(2)
#if defined(MSB_STUDY_REPEAT)
    run(k, MSB_STUDY_LIMIT);
#else
    run(k);
#endif
    if (fault()) return action == 155 ? k.result : 0;   // 0 if the play raised; what was computed before the turn was passed on otherwise
    if (action != 155) k.result = (pl_base(remote()) <= 0 ? 1 : 0) | (have_winner() ? 2 : 0);
    if (m.ld8(H_RNGOVER)) set_fault(FAULT_RNG_OVERRUN);
    return k.result;
  }

  // The handlers of the remaining frames, and run(): the only loop of a step.
  MSB_HD MSB_INL void h_ctx_leave(Wk& k, const uint32_t hdr) {
    ctx_leave(hdr_a(hdr));
    k.sp--;
  }
  MSB_HD MSB_INL void h_destroy_tail(Wk& k, const uint32_t) {
    recalc_front_after_destroy();
    k.sp--;
  }
  MSB_HD MSB_INL void h_cmd_tail(Wk& k, const uint32_t hdr) {
    e_set_flag(hdr_a(hdr), EF_FF, hdr_b(hdr) != 0);
    k.sp--;
  }
  MSB_HD MSB_INL void wk_dispatch(Wk& k, int fn, const uint32_t hdr) {
    switch (fn) {
      case F_MOVE: {
        MSB_SCOPE(PS_MOVE);
        h_move(k, hdr);
      } break;
      case F_RUNAB: {
        MSB_SCOPE(PS_RUN_ABILITY);
        h_runab(k, hdr);
      } break;
      case F_CTXLEAVE: h_ctx_leave(k, hdr); break;
      case F_DESTROY_TAIL: h_destroy_tail(k, hdr); break;
      case F_CMD_TAIL: h_cmd_tail(k, hdr); break;
      // the rare frames run out of line (MSB_A_RARE): their code stays out of the registers and the instruction stream
      // of the two hot handlers
      case F_EACH: k.sp = h_each_out(k.sp, k.base - k.seg, hdr); break;
      case F_AFTER: k.sp = h_after_out(k.sp, k.base - k.seg, hdr); break;
      case F_TURN: k.sp = h_turn_out(k.sp, k.base - k.seg, hdr); break;
      case F_EVICTED: h_evicted(k, hdr); break;
      default: set_fault(FAULT_UNSUPPORTED); break;   // not a frame: cannot happen
    }
  }
  // Out-of-line forms: the stack pointer goes in and comes back by value (a Wk passed by reference would live in memory);
  // `deep` = words evicted - eviction marks, all wk_reserve needs.
  MSB_HD MSB_A_RARE int h_each_out(int sp, int deep, uint32_t hdr) {
    MSB_SCOPE(PS_FORCE_ATTACK);
    Wk k{sp, 0, deep, 0};
    h_each(k, hdr);
    return k.sp;
  }
  MSB_HD MSB_A_RARE int h_after_out(int sp, int deep, uint32_t hdr) {
    Wk k{sp, 0, deep, 0};
    h_after(k, hdr);
    return k.sp;
  }
  MSB_HD MSB_A_RARE int h_turn_out(int sp, int deep, uint32_t hdr) {
    MSB_SCOPE(PS_NEXT_TURN);
    Wk k{sp, 0, deep, 0};
    h_turn(k, hdr);
    return k.sp;
  }
  MSB_HD MSB_A_RARE int next_turn_out(int sp) {
    MSB_SCOPE(PS_FLIP);
    Wk k{sp, 0, 0, 0};
    call_next_turn(k);
    return k.sp;
  }
  // Run the work stack until it is empty.  On the device the lanes of a wave are grouped by the function of their top
  // frame first (a "waterfall": take the first waiting lane's function as a wave-uniform value, serve the lanes
  // holding it, repeat), so the switch runs on a scalar and lanes that are in the same function -- whatever path of
  // calls took them there -- execute it together.
#if defined(MSB_STUDY_REPEAT)
  MSB_HD MSB_INL void run(Wk& k, int rounds = 1 << 30) {
#define MSB_STUDY_ROUND && rounds-- > 0
#else
  MSB_HD MSB_INL void run(Wk& k) {
#define MSB_STUDY_ROUND
#endif
    MSB_SCOPE(PS_COMMAND);   // profiling build: the whole loop (the handlers' own scopes are inside it)
    while (k.sp > 0 && !fault() MSB_STUDY_ROUND) {
      uint32_t hdr = m.sk_ld(k.sp - 1);
      const int fn = hdr_fn(hdr);
      if (M::SKW < SK_CAP && k.sp + SK_NEED > M::SKW && fn != F_EVICTED) wk_evict(k, fn);
#if defined(__HIP_DEVICE_COMPILE__)
      int fv = fn;
      asm volatile("" : "+v"(fv));   // an opaque copy: under `fn == f0` the compiler would switch on the vector `fn` again
      for (unsigned long long todo = __ballot(1); todo;) {
        const int leader = __builtin_ctzll(todo);
        const int f0 = __builtin_amdgcn_readlane(fn, leader), f1 = __builtin_amdgcn_readlane(fv, leader);
        const bool mine = fn == f0;
        todo &= ~__ballot(mine);
        if (mine) {
          // opaque copies: whatever a handler derives from the header or the stack pointer is computed inside the
          // branch that runs it (hoisted out of this loop, the sub-expressions of ALL handlers would be computed for
          // every frame -- and spilled)
          uint32_t hv = hdr;
          asm volatile("" : "+v"(hv), "+v"(k.sp));
          wk_dispatch(k, f1, hv);
        }
      }
#else
#if defined(MSB_COUNT_FRAMES)
      msb_frame_count[fn]++;
      msb_frame_count[0]++;
      if (k.base + k.sp > msb_frame_count[15]) msb_frame_count[15] = k.base + k.sp;
      msb_frame_count[16 + (k.sp < 47 ? k.sp / 4 : 11)]++;
      if (fn == F_MOVE) msb_frame_count[32 + (hdr_st(hdr) & 15)]++;
      if (fn == F_RUNAB) msb_frame_count[48 + (hdr_st(hdr) & 1)]++;
#endif
      wk_dispatch(k, fn, hdr);
#endif
    }
  }


  // Stormbound.expert_action, games/stormbound.py:563-637: the reference's scripted opponent.  Draws from
  // the GAME's stream (self.random).  May return PASS while plays are still legal -- that is how it ends a turn.
  MSB_HD MSB_NOINLINE int expert_action() {
    msb_u64x4 lm = legal_mask_v();
    int lo = local();
    int hn = pl_hand_n(lo), mana = pl_mana(lo);
    if ((lm[2] >> (148 - 128)) & 0xf) {   // any REPLACE offered
      int max_cost = -1;
      for (int i = 0; i < hn; i++) max_cost = hand_cost(lo, i) > max_cost ? hand_cost(lo, i) : max_cost;
      if (max_cost > mana) {
        PList idx;
        idx.clear();
        for (int i = 0; i < hn; i++)
          if (hand_cost(lo, i) == max_cost) idx.push_raw(i);
        return 148 + idx.get(choice_index(idx.n()));
      }
    }
    PList playable;
    playable.clear();
    for (int i = 0; i < 4; i++) {
      // any x in [16i, 16i+15] or [21i+64, 21i+84]
      bool any = false;
      for (int a = 16 * i; a <= 16 * i + 15 && !any; a++) any = (lm[0] >> a) & 1;
      for (int a = 21 * i + 64; a <= 21 * i + 84 && !any; a++) any = a < 128 ? ((lm[1] >> (a - 64)) & 1) : ((lm[2] >> (a - 128)) & 1);
      if (any) playable.push_raw(i);
    }
    if (playable.n() > 0) {
      // cards whose cost equals the current mana, else the cheapest ones; random among them
      bool exact = false;
      int min_cost = 1 << 20;
      for (int k = 0; k < playable.n(); k++) {
        int c = hand_cost(lo, playable.get(k));
        if (c == mana) exact = true;
        if (c < min_cost) min_cost = c;
      }
      int want = exact ? mana : min_cost;
      PList pick;
      pick.clear();
      for (int k = 0; k < playable.n(); k++)
        if (hand_cost(lo, playable.get(k)) == want) pick.push_raw(k);
      int index = playable.get(pick.get(choice_index(pick.n())));
      int card = hand_card(lo, index);
      const CardInfo& ci = g_cards[card];
      PList enemies = get_targets(cp(), mk_tgt(TK_UNIT, TS_ENEMY), PK_NONE);
      PList bbe;   // base_bordering_enemies
      bbe.clear();
      for (int k = 0; k < enemies.n(); k++)
        if (enemies.at(k).y == 4) bbe.push_raw(enemies.get(k));
      if (ci.kind == KIND_SPELL) {
        if (!ci.tgt.has) return 64 + 21 * index;
        PList t = get_targets(cp(), mk_tgt(ci.tgt), PK_NONE);
        if (t.n() == 0) {
          set_fault(FAULT_PY_EXCEPTION);   // random.choice([]) raises
          return 155;
        }
        P p = choice_point(t);
        return p_valid(p) ? 65 + 21 * index + (4 - p.y) * 4 + p.x : 155;
      } else if (ci.kind == KIND_UNIT && bbe.n() > 0) {
        PList cand;
        cand.clear();
        for (int k = 0; k < bbe.n(); k++) {
          P en = bbe.at(k);
          if (en.x > 0 && at(P{en.x - 1, en.y}) == AT_NONE)
            cand.push(P{en.x - 1, en.y});
          else if (en.x < 3 && at(P{en.x + 1, en.y}) == AT_NONE)
            cand.push(P{en.x + 1, en.y});
        }
        if (cand.n() > 0) {
          P p = choice_point(cand);
          return p.y >= 1 ? 16 * index + (4 - p.y) * 4 + p.x : 155;
        }
      } else {
        int fl = pl_front(lo);
        PList cand;
        cand.clear();
        for (int x = 0; x < 4; x++)
          if (at(P{x, fl}) == AT_NONE) cand.push(P{x, fl});
        for (int k = 0; k < enemies.n(); k++) {
          P en = enemies.at(k);
          if (en.x > 0 && en.y >= fl && at(P{en.x - 1, en.y}) == AT_NONE)
            cand.push(P{en.x - 1, en.y});
          else if (en.x < 3 && en.y >= fl && at(P{en.x + 1, en.y}) == AT_NONE)
            cand.push(P{en.x + 1, en.y});
          else if (en.y < 4 && en.y + 1 >= fl && at(P{en.x, en.y + 1}) == AT_NONE)
            cand.push(P{en.x, en.y + 1});
        }
        if (cand.n() > 0) {
          P p = choice_point(cand);
          return p.y >= 1 ? 16 * index + (4 - p.y) * 4 + p.x : 155;
        }
      }
    }
    if (m.ld8(H_RNGOVER)) set_fault(FAULT_RNG_OVERRUN);
    return 155;
  }

  // Game construction: Stormbound.__init__ / Player.__init__ (games/stormbound.py:293-304,
  // player.py:13-37).  deck0/deck1: 12 card indices in constructor order.
  MSB_HD MSB_NOINLINE void init_game(const uint8_t* deck0, const uint8_t* deck1, int faction0, int faction1, uint32_t seed = 0) {
    uint64_t rc = m.ld64(H_RNGCUR), rn = m.ld64(H_RNGNXT);
    uint32_t rp = rng_pos();
    for (int w = 0; w < STATE_WORDS; w++) m.st32(4 * w, 0);
    m.st64(H_RNGCUR, rc);
    m.st64(H_RNGNXT, rn);
    m.st16(H_RNGPOS, (int)rp);
    for (int t = 0; t < 20; t++) board_put(t, SLOT_NONE);
    for (int e = 0; e < NUM_ENT; e++) m.st8g(eg(e), EO_CARD, CARD_NONE);
    if (REM_LISTS) {
      for (int e = 0; e < NUM_ENT; e++) m.st8(E_REM + e, REM_NONE);
      set_seed(seed);
    }
    for (int i = 0; i < 4; i++) {
      m.st8(H_HIST + 2 * i, 0xff);
      m.st8(H_HIST + 2 * i + 1, 0xff);
    }
    m.st8(H_PHASE, PH_PLAY);
    for (int o = 0; o < 2; o++) {
      const uint8_t* deck = o == 0 ? deck0 : deck1;
      m.st16(pl(o, P_BASE), 20);
      m.st16(pl(o, P_MAXMANA), o == 0 ? 3 : 4);
      m.st16(pl(o, P_MANA), o == 0 ? 3 : 4);
      m.st8(pl(o, P_FRONT), o == 0 ? 4 : 0);
      m.st8(pl(o, P_FLAGS), 3);
      m.st8(pl(o, P_FACTION), o == 0 ? faction0 : faction1);
      uint8_t d[DECK_SIZE];
      for (int i = 0; i < DECK_SIZE; i++) {
        d[i] = deck[i];
        if (g_cards[d[i]].int_id < 0) m.st8(H_OBSFAULT, m.ld8(H_OBSFAULT) | GF_OBSFAULT);
      }
      for (int i = DECK_SIZE - 1; i >= 1; i--) {  // random.shuffle(self.deck)
        int j = (int)rng_interval((uint32_t)i);
        uint8_t tmp = d[i];
        d[i] = d[j];
        d[j] = tmp;
      }
      for (int i = 0; i < DECK_SIZE; i++) {   // weights 1, f(1), f(f(1)), ... (player.py:29-31): age = position
#if defined(MSB_EXT) && MSB_EXT
        m.st8(pl(o, P_DECK + i), i);   // object i sits at deck position i
#endif
        m.st8(pl(o, P_DECK_N), i + 1);
        int r = deck_ref(o, i);
        m.st8(r, d[i]);
        m.st8(r + 1, g_cards[d[i]].cost);
        m.st8(r + 2, (g_cards[d[i]].ff ? CF_FF : 0) | CF_XBASE | (g_cards[d[i]].kind == KIND_SPELL ? CF_SPELL : 0));
        m.st8(r + 3, g_cards[d[i]].strength);
        set_deck_age(o, i, i);
      }
      m.st8(pl(o, P_DECK_N), DECK_SIZE);
      fill_hand(o);
    }
    if (m.ld8(H_RNGOVER)) set_fault(FAULT_RNG_OVERRUN);
  }

  // ------------------------------------------------------------------------------------------
  // Card abilities (cards/*.py).  Dispatch by card index = the device-side opcode table.
  // ------------------------------------------------------------------------------------------
#include "abilities.inc"
#include "observe.inc"
#include "scenario.inc"
};

}  // namespace msb
