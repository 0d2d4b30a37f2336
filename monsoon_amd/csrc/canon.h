// Canonical state record v1: the slot-free, position-ordered serialisation that bit-exactness is
// judged on (same byte layout as oracle/pyref/harness.py::canon builds from the reference's
// objects).  Used by monsoon_state_export (product) and by the oracle.
#pragma once
// (included behind a rules core: the product's rules.h or, in the oracle, oracle/recursive/rules.h)

namespace msb {

constexpr int CANON_MAX = 4 + 8 + 2 * (12 + 3 * HAND_CAP + 11 * DECK_CAP) + 20 * 11 + 4;
static_assert(CANON_MAX <= 2048, "monsoon_state_export hands out at most 2048 bytes (include/monsoon.h)");

template <class M>
MSB_HD inline int canon_record(const Engine<M>& g, uint32_t next_u32, uint8_t* out) {
  int n = 0;
  auto p8 = [&](int v) { out[n++] = (uint8_t)v; };
  auto p16 = [&](int v) { out[n++] = (uint8_t)(v & 0xff); out[n++] = (uint8_t)((v >> 8) & 0xff); };
  p8(g.m.ld8(H_TOPLAY));
  p8(g.local());
  p8(g.m.ld8(H_HIST_N));
  p8(0);
  int hn = g.m.ld8(H_HIST_N);
  for (int i = 0; i < 4; i++) {
    int k = i - (4 - hn);
    if (k >= 0) {
      p8(g.m.ld8(H_HIST + 2 * k));
      p8(g.m.ld8(H_HIST + 2 * k + 1));
    } else {
      p8(0xff);
      p8(0xff);
    }
  }
  for (int o = 0; o < 2; o++) {
    p16(g.pl_base(o));
    p16(g.pl_mana(o));
    p16(g.pl_maxmana(o));
    p8(g.pl_front(o));
    p8(g.m.ld8(g.pl(o, P_FLAGS)) & 3);
    p8(g.m.ld8(g.pl(o, P_FACTION)));
    p8(g.pl_hand_n(o));
    p8(g.pl_deck_n(o));
    p8(0);
    for (int i = 0; i < g.pl_hand_n(o); i++) {
      p8(g.hand_card(o, i));
      p8(g.hand_cost(o, i));
      p8(g.hand_flags(o, i) & (CF_SINGLE_USE | CF_FF));
    }
    for (int i = 0; i < g.pl_deck_n(o); i++) {
      p8(g.deck_card(o, i));
      p8(g.deck_cost(o, i));
      p8(g.deck_flags(o, i) & (CF_SINGLE_USE | CF_FF));
      union { double d; uint8_t b[8]; } u;
      u.d = g.deck_w(o, i);
      for (int k = 0; k < 8; k++) p8(u.b[k]);
    }
  }
  for (int t = 0; t < 20; t++) {
    int s = g.board_at(t);
    if (s == SLOT_NONE) {
      p8(0xff);
      continue;
    }
    p8(g.e_card(s));
    p8(g.e_flags(s) & (EF_OWNER | EF_FF));
    p16(g.e_str(s));
    p8(g.card_is_unit(g.e_card(s)) ? g.e_mov(s) : 0);
    p8(g.m.ld8(OFF_ENT + ENT_SIZE * s + EO_POS));
    for (int k = 0; k < 5; k++) p8(g.e_st(s, k));
  }
  for (int k = 0; k < 4; k++) p8((next_u32 >> (8 * k)) & 0xff);
  return n;
}

MSB_HD inline uint64_t fnv1a64(const uint8_t* p, int n) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (int i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
  return h;
}

}  // namespace msb
