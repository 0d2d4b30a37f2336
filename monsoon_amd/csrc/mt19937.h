// MT19937 + numpy legacy RandomState draw semantics.
//
// The reference draws every random decision of a game from one
// numpy.random.RandomState (games/stormbound.py:294, player.py:28,49, unit.py:95 and the
// cards).  numpy is a third-party dependency absent from /root/reference (requirements.txt:3,
// unpinned; the build container has 2.2.6); its legacy stream is frozen by NEP 19.  This file
// restates the published algorithm:
//   init_genrand(seed)           mt[0]=seed; mt[i]=1812433253*(mt[i-1]^(mt[i-1]>>30))+i
//   genrand                      standard twist (N=624, M=397, 0x9908b0df) + tempering
//   random_sample()              (a>>5, b>>6) -> (a*2^26+b)/2^53
//   interval(max)                0 draws if max==0; mask = 2^k-1 >= max; redraw (u32 & mask) until <= max
//   randint(lo,hi)               lo + interval(hi-1-lo)
//   choice(list)                 list[randint(0,len)]
//   shuffle(list)                for i=n-1..1: j=interval(i); swap(i,j)
//   choice(a,size=1,p)           cdf=cumsum(p); cdf/=cdf[-1]; u=random_sample(); searchsorted(cdf,u,'right')
// Pinned by tests/golden/rng_kat.npz, generated from numpy itself by oracle/pyref/gen_golden.py.
#pragma once
#include "msb_base.h"

namespace msb {

constexpr int MT_N = 624;
constexpr int MT_M = 397;

MSB_HD MSB_INL uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

MSB_HD MSB_INL uint32_t mt_mix(uint32_t a, uint32_t b) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Serial reference forms (host oracle, and single-lane device use in tests).
MSB_HD inline void mt_seed(uint32_t* mt, uint32_t seed) {
  mt[0] = seed;
  for (int i = 1; i < MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

MSB_HD inline void mt_twist(uint32_t* mt) {
  int k;
  for (k = 0; k < MT_N - MT_M; k++) mt[k] = mt[k + MT_M] ^ mt_mix(mt[k], mt[k + 1]);
  for (; k < MT_N - 1; k++) mt[k] = mt[k + (MT_M - MT_N)] ^ mt_mix(mt[k], mt[k + 1]);
  mt[MT_N - 1] = mt[MT_M - 1] ^ mt_mix(mt[MT_N - 1], mt[0]);
}

// The stream a game step sees is a read-only window over two consecutive blocks of TEMPERED outputs
// (current + next) with a cursor; the window and the cursor are fields of the state record (state.h:
// H_RNGCUR, H_RNGNXT, H_RNGPOS) and the draw functions are Engine methods (rules.h: rng_*).  A look-ahead
// step keeps a private cursor (the reference's copy.deepcopy clones the stream,
// evo/game_adapter.py:284), so many candidate steps of one decision read the same window.
#if defined(__HIPCC__)
#define MSB_RNG_PTR __attribute__((address_space(1))) const uint32_t*   // global_load, not flat_load
#else
#define MSB_RNG_PTR const uint32_t*
#endif

}  // namespace msb
