// sw_mt19937.cpp -- host-side probe stream of the engine (no GPU needed).
//
// The reference draws its Rademacher probes with np.random.randint(2, size=N) from the global
// legacy NumPy generator (utils.py:213-216, 255-258; seeded at stoch_trace.py:103,288).  That is
// MT19937 seeded by init_genrand(seed), one 32-bit output per entry, entry = output & 1, and the
// stream continues across calls.  This file produces the same stream so probe
// batches can be generated (and sharded by stream position across ranks) without NumPy.
#include "../../include/schwinger_hip.h"

#include <cstdlib>

struct sw_mt19937 {
  uint32_t mt[624];
  int idx;
};

static void mt_refill(sw_mt19937* g) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
  uint32_t* mt = g->mt;
  int k = 0;
  for (; k < 624 - 397; ++k) {
    const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);
    mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
  }
  for (; k < 623; ++k) {
    const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);
    mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
  }
  const uint32_t y = (mt[623] & UPPER) | (mt[0] & LOWER);
  mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
  g->idx = 0;
}

static inline uint32_t mt_next(sw_mt19937* g) {
  if (g->idx >= 624) mt_refill(g);
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

extern "C" {

sw_mt19937* sw_mt_create(uint32_t seed) {
  sw_mt19937* g = (sw_mt19937*)std::malloc(sizeof(sw_mt19937));
  if (!g) return nullptr;
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
  return g;
}

void sw_mt_destroy(sw_mt19937* g) { std::free(g); }

void sw_mt_skip(sw_mt19937* g, uint64_t ndraws) {
  // whole-state strides first, then the remainder
  while (ndraws > 0) {
    if (g->idx >= 624) mt_refill(g);
    const uint64_t avail = 624 - g->idx;
    const uint64_t take = ndraws < avail ? ndraws : avail;
    g->idx += (int)take;
    ndraws -= take;
  }
}

void sw_mt_raw(sw_mt19937* g, uint64_t n, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = mt_next(g);
}

void sw_mt_rademacher(sw_mt19937* g, uint64_t n, int8_t* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = (int8_t)(2 * (int)(mt_next(g) & 1u) - 1);
}

}  // extern "C"
