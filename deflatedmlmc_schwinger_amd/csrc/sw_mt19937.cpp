// sw_mt19937.cpp -- host side of the engine's probe stream (no GPU needed).
//
// The reference draws its Rademacher probes with np.random.randint(2, size=N) from the global
// legacy NumPy generator (utils.py:213-216, 255-258; seeded at stoch_trace.py:103,288).  That is
// MT19937 seeded by init_genrand(seed), one 32-bit output per entry, entry = output & 1, and the
// stream continues across calls.  This file
//   * produces the same stream word by word (the checker of the device generator),
//   * and supplies what the DEVICE generator (k_mt_jump / k_mt_generate in sw_kernels.hpp) needs to
//     start anywhere in the stream without walking there: jump polynomials g_J(x) = x^J mod phi(x)
//     over GF(2), phi = characteristic polynomial of the MT19937 word recurrence (Haramoto,
//     Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number
//     generators", 2008).  With w_t the raw (untempered) word sequence, the 624-word window at
//     position p+J is  W'[k] = XOR_{i : g_i = 1} w[p+i+k],  a GF(2) convolution that is evaluated
//     in parallel on the GPU (and below on the host).
// phi is not tabulated: it is recovered once per process from the generator's own output with
// Berlekamp-Massey (minimal polynomial of one output bit; phi is primitive, so this is phi).
#include "../../include/schwinger_hip.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

struct sw_mt19937 {
  uint32_t mt[624];
  int idx;
};

namespace {

const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;

inline uint32_t twist(uint32_t a, uint32_t b) {
  const uint32_t y = (a & UPPER) | (b & LOWER);
  return (y >> 1) ^ ((y & 1u) ? MAG : 0u);
}

void mt_refill(sw_mt19937* g) {
  uint32_t* mt = g->mt;
  int k = 0;
  for (; k < 624 - 397; ++k) mt[k] = mt[k + 397] ^ twist(mt[k], mt[k + 1]);
  for (; k < 623; ++k) mt[k] = mt[k + (397 - 624)] ^ twist(mt[k], mt[k + 1]);
  mt[623] = mt[396] ^ twist(mt[623], mt[0]);
  g->idx = 0;
}

inline uint32_t temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline uint32_t mt_next(sw_mt19937* g) {
  if (g->idx >= 624) mt_refill(g);
  return temper(g->mt[g->idx++]);
}

void mt_seed(sw_mt19937* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

// ---------------------------------------------------------------------------------------------
// GF(2)[x] modulo phi.  Bit i of a word array = coefficient of x^i.
// ---------------------------------------------------------------------------------------------
const int DEG = 19937;
const int PW = 312;          // u64 words of a reduced polynomial (19968 bits)
const int SW = PW + 2;       // words of a polynomial shifted left by up to 63 bits
const int PRODW = 2 * PW + 2;

struct Field {
  bool ok = false;
  uint64_t phi[SW];
  std::vector<uint64_t> phis;                      // [64][SW]: phi << r
  std::map<uint64_t, std::vector<uint64_t>> cache; // J -> x^J mod phi (PW words)
};
Field g_field;
std::mutex g_field_mutex;

inline int parity64(uint64_t v) { return __builtin_parityll(v); }

void shifted_copies(const uint64_t* p, int nw, std::vector<uint64_t>& out) {
  // out[r*SW + j] = (p << r) word j, r = 0..63
  out.assign((size_t)64 * SW, 0);
  for (int r = 0; r < 64; ++r) {
    uint64_t* o = &out[(size_t)r * SW];
    uint64_t carry = 0;
    for (int j = 0; j < nw; ++j) {
      o[j] = r ? ((p[j] << r) | carry) : p[j];
      carry = r ? (p[j] >> (64 - r)) : 0;
    }
    if (nw < SW) o[nw] = carry;
  }
}

// Berlekamp-Massey over GF(2) on N bits of one output-bit sequence of the generator
bool build_field(Field& F) {
  const int N = 2 * DEG + 64;
  sw_mt19937 g;
  mt_seed(&g, 5489u);
  std::vector<uint8_t> s(N);
  for (int t = 0; t < N; ++t) {
    if (g.idx >= 624) mt_refill(&g);
    s[t] = (uint8_t)(g.mt[g.idx++] & 1u);   // bit 0 of the raw word: a linear functional of the state
  }
  const int NW = SW;
  std::vector<uint64_t> Cc(NW, 0), Bb(NW, 0), Tt(NW, 0), rev(NW, 0);
  Cc[0] = 1;
  Bb[0] = 1;
  int L = 0, m = 1;
  for (int n = 0; n < N; ++n) {
    uint64_t carry = s[n];
    const int wtop = std::min(NW, n / 64 + 2);
    for (int w = 0; w < wtop; ++w) {
      const uint64_t nc = rev[w] >> 63;
      rev[w] = (rev[w] << 1) | carry;
      carry = nc;
    }
    uint64_t acc = 0;
    const int wl = std::min(NW, L / 64 + 1);
    for (int w = 0; w < wl; ++w) acc ^= Cc[w] & rev[w];
    if (!parity64(acc)) {
      ++m;
      continue;
    }
    const bool grow = (2 * L <= n);
    if (grow) Tt = Cc;
    // C ^= B << m
    const int ws = m / 64, bs = m % 64;
    for (int w = NW - 1; w >= ws; --w) {
      uint64_t v = Bb[w - ws] << bs;
      if (bs && w - ws - 1 >= 0) v |= Bb[w - ws - 1] >> (64 - bs);
      Cc[w] ^= v;
    }
    if (grow) {
      L = n + 1 - L;
      Bb = Tt;
      m = 1;
    } else {
      ++m;
    }
    if (L > DEG) return false;
  }
  if (L != DEG) return false;
  // characteristic polynomial phi_j = C_{L-j}
  std::memset(F.phi, 0, sizeof F.phi);
  for (int j = 0; j <= DEG; ++j) {
    const int i = DEG - j;
    if ((Cc[i / 64] >> (i % 64)) & 1ull) F.phi[j / 64] |= 1ull << (j % 64);
  }
  if (!((F.phi[DEG / 64] >> (DEG % 64)) & 1ull) || !(F.phi[0] & 1ull)) return false;
  shifted_copies(F.phi, PW, F.phis);
  F.ok = true;
  return true;
}

// prod (PRODW words) -> reduced in place to PW words
void reduce(const Field& F, uint64_t* prod) {
  for (int i = 2 * DEG; i >= DEG; --i) {
    if (!((prod[i / 64] >> (i % 64)) & 1ull)) continue;
    const int d = i - DEG, wd = d / 64, r = d % 64;
    const uint64_t* ps = &F.phis[(size_t)r * SW];
    uint64_t* dst = prod + wd;
    for (int j = 0; j < PW + 1; ++j) dst[j] ^= ps[j];
  }
}

void mulmod(const Field& F, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  std::vector<uint64_t> bs;
  shifted_copies(b, PW, bs);
  std::vector<uint64_t> prod(PRODW + 2, 0);
  for (int w = 0; w < PW; ++w) {
    uint64_t bits = a[w];
    while (bits) {
      const int r = __builtin_ctzll(bits);
      bits &= bits - 1;
      const uint64_t* src = &bs[(size_t)r * SW];
      uint64_t* dst = &prod[w];
      for (int j = 0; j < PW + 1; ++j) dst[j] ^= src[j];
    }
  }
  reduce(F, prod.data());
  std::memcpy(out, prod.data(), PW * sizeof(uint64_t));
}

// x^J mod phi (cached); returns nullptr when the field could not be built
const uint64_t* jump_poly(uint64_t J) {
  std::lock_guard<std::mutex> lock(g_field_mutex);
  Field& F = g_field;
  if (!F.ok && !build_field(F)) return nullptr;
  auto it = F.cache.find(J);
  if (it != F.cache.end()) return it->second.data();
  std::vector<uint64_t> r(PW, 0), t(PW, 0);
  r[0] = 1;
  // chain: a cached x^(J - J0) * x^J0 is cheaper than 40 squarings when a common stride exists
  // (the device generator asks for J0, 2 J0, 3 J0, ...)
  bool done = false;
  for (auto& kv : F.cache) {
    const uint64_t J0 = kv.first;
    if (J0 == 0 || J0 >= J) continue;
    auto jt = F.cache.find(J - J0);
    if (jt != F.cache.end()) {
      mulmod(F, kv.second.data(), jt->second.data(), r.data());
      done = true;
      break;
    }
  }
  if (!done) {
    int top = 63;
    while (top > 0 && !((J >> top) & 1ull)) --top;
    for (int b = top; b >= 0; --b) {
      mulmod(F, r.data(), r.data(), t.data());
      r.swap(t);
      if ((J >> b) & 1ull) {
        // r *= x
        uint64_t carry = 0;
        for (int w = 0; w < PW; ++w) {
          const uint64_t nc = r[w] >> 63;
          r[w] = (r[w] << 1) | carry;
          carry = nc;
        }
        if ((r[DEG / 64] >> (DEG % 64)) & 1ull)
          for (int w = 0; w < PW; ++w) r[w] ^= F.phi[w];
      }
    }
  }
  auto ins = F.cache.emplace(J, r);
  return ins.first->second.data();
}

// window at position p -> window at position p + J
bool window_jump(const uint32_t* win, uint64_t J, uint32_t* out) {
  if (J == 0) {
    std::memmove(out, win, 624 * sizeof(uint32_t));
    return true;
  }
  const uint64_t* g = jump_poly(J);
  if (!g) return false;
  const int NWORDS = DEG + 624;   // w[i + k], i <= DEG-1, k <= 623
  std::vector<uint32_t> w(NWORDS + 8);
  std::memcpy(w.data(), win, 624 * sizeof(uint32_t));
  for (int t = 0; t + 624 < NWORDS; ++t) w[t + 624] = w[t + 397] ^ twist(w[t], w[t + 1]);
  uint32_t acc[624];
  std::memset(acc, 0, sizeof acc);
  for (int wi = 0; wi < PW; ++wi) {
    uint64_t bits = g[wi];
    while (bits) {
      const int i = wi * 64 + __builtin_ctzll(bits);
      bits &= bits - 1;
      const uint32_t* src = &w[i];
      for (int k = 0; k < 624; ++k) acc[k] ^= src[k];
    }
  }
  std::memcpy(out, acc, sizeof acc);
  return true;
}

}  // namespace

extern "C" {

sw_mt19937* sw_mt_create(uint32_t seed) {
  sw_mt19937* g = (sw_mt19937*)std::malloc(sizeof(sw_mt19937));
  if (!g) return nullptr;
  mt_seed(g, seed);
  return g;
}

sw_mt19937* sw_mt_from_state(const uint32_t* key, int pos) {
  if (!key || pos < 0 || pos > 624) return nullptr;
  sw_mt19937* g = (sw_mt19937*)std::malloc(sizeof(sw_mt19937));
  if (!g) return nullptr;
  std::memcpy(g->mt, key, sizeof g->mt);
  g->idx = pos;
  return g;
}

void sw_mt_get_state(const sw_mt19937* g, uint32_t* key, int* pos) {
  if (key) std::memcpy(key, g->mt, sizeof g->mt);
  if (pos) *pos = g->idx;
}

void sw_mt_destroy(sw_mt19937* g) { std::free(g); }

void sw_mt_skip(sw_mt19937* g, uint64_t ndraws) {
  // sequential walk, one state refill per 624 draws: the checker of sw_mt_jump
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

void sw_mt_z4(sw_mt19937* g, uint64_t n, int8_t* out) {
  static const int8_t code[4] = {1, 2, -1, -2};
  for (uint64_t i = 0; i < n; ++i) out[i] = code[mt_next(g) & 3u];
}

void sw_mt_window(sw_mt19937* g, uint32_t* out) {
  if (g->idx >= 624) mt_refill(g);
  if (g->idx == 0) {
    std::memcpy(out, g->mt, sizeof g->mt);
    return;
  }
  sw_mt19937 nx = *g;
  mt_refill(&nx);
  const int i = g->idx;
  std::memcpy(out, g->mt + i, (624 - i) * sizeof(uint32_t));
  std::memcpy(out + (624 - i), nx.mt, i * sizeof(uint32_t));
}

int sw_mt_jump_poly(uint64_t ndraws, uint32_t* poly) {
  const uint64_t* g = jump_poly(ndraws);
  if (!g) return 1;
  std::memcpy(poly, g, 624 * sizeof(uint32_t));
  return 0;
}

int sw_mt_window_jump(const uint32_t* win, uint64_t ndraws, uint32_t* out) {
  return window_jump(win, ndraws, out) ? 0 : 1;
}

int sw_mt_jump(sw_mt19937* g, uint64_t ndraws) {
  // same (key, pos) representation NumPy would hold after drawing `ndraws` more words, except
  // that a position on a block boundary is stored as (next block, 0) instead of (block, 624)
  if (g->idx >= 624) mt_refill(g);
  const uint64_t target = (uint64_t)g->idx + ndraws;
  const uint64_t nblocks = target / 624;
  if (nblocks > 0) {
    uint32_t out[624];
    if (!window_jump(g->mt, nblocks * 624, out)) return 1;
    std::memcpy(g->mt, out, sizeof out);
  }
  g->idx = (int)(target % 624);
  return 0;
}

}  // extern "C"
