// host_sanitize.cpp -- driver of `make sanitize`: the engine's host-side code (CSR packers of
// sw_pack.hpp, the MT19937 stream and its GF(2) jump machinery of sw_mt19937.cpp) built with
// -fsanitize=address,undefined and exercised on small random inputs.  No GPU, no HIP.
// Exit code 0 = all checks passed and the sanitizers saw nothing.
#include "../../include/schwinger_hip.h"
#include "sw_pack.hpp"
#include "sw_poly.hpp"

#include <cstdio>
#include <cstring>
#include <random>

typedef std::complex<double> cd;

static int fails = 0;
#define CHECK(cond, ...)                 \
  do {                                   \
    if (!(cond)) {                       \
      std::printf("FAIL: " __VA_ARGS__); \
      std::printf("\n");                 \
      ++fails;                           \
    }                                    \
  } while (0)

struct Csr {
  int nrows, ncols;
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<cd> data;
};

static Csr random_csr(int nrows, int ncols, int per_row, std::mt19937& rng, bool blocky) {
  Csr A;
  A.nrows = nrows;
  A.ncols = ncols;
  A.indptr.push_back(0);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int r = 0; r < nrows; ++r) {
    std::vector<int> cols;
    if (blocky) {
      const int base = ((r / 16) * 16) % ncols;            // block-row structure: shared columns
      for (int k = 0; k < per_row; ++k) cols.push_back((base + 4 * (k % 5) * 16 + k) % ncols);
    } else {
      const int cnt = (int)(rng() % (unsigned)(per_row + 1));   // ragged, possibly empty rows
      for (int k = 0; k < cnt; ++k) cols.push_back((int)(rng() % (unsigned)ncols));
    }
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    for (int c : cols) {
      A.indices.push_back(c);
      A.data.push_back(cd(u(rng), u(rng)));
    }
    A.indptr.push_back((int64_t)A.indices.size());
  }
  return A;
}

static std::vector<cd> csr_apply(const Csr& A, const std::vector<cd>& x) {
  std::vector<cd> y(A.nrows, cd(0, 0));
  for (int r = 0; r < A.nrows; ++r)
    for (int64_t q = A.indptr[r]; q < A.indptr[r + 1]; ++q) y[r] += A.data[q] * x[A.indices[q]];
  return y;
}

static void test_ell(std::mt19937& rng) {
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  const int shapes[][3] = {{64, 64, 9}, {48, 16, 4}, {16, 48, 12}, {30, 7, 3}, {1, 1, 1}, {32, 32, 0}};
  for (auto& sh : shapes) {
    Csr A = random_csr(sh[0], sh[1], sh[2], rng, false);
    for (int variant = 0; variant < 3; ++variant) {
      std::vector<int> rows_int, colmap;
      if (variant >= 1) {   // row permutation (internal row r holds natural row rows_int[r])
        rows_int.resize(A.nrows);
        for (int i = 0; i < A.nrows; ++i) rows_int[i] = i;
        std::shuffle(rows_int.begin(), rows_int.end(), rng);
      }
      if (variant == 2) {
        colmap.resize(A.ncols);
        for (int i = 0; i < A.ncols; ++i) colmap[i] = i;
        std::shuffle(colmap.begin(), colmap.end(), rng);
      }
      swp::EllHost e;
      std::string err;
      const int rc = swp::ell_pack(e, err, A.nrows, A.ncols, A.indptr.data(), A.indices.data(),
                                   A.data.data(), rows_int, colmap, 0);
      CHECK(rc == 0, "ell_pack rc %d (%s)", rc, err.c_str());
      if (rc) continue;
      std::vector<cd> x(A.ncols), xi(A.ncols);
      for (auto& v : x) v = cd(u(rng), u(rng));
      for (int c = 0; c < A.ncols; ++c) xi[colmap.empty() ? c : colmap[c]] = x[c];
      const std::vector<cd> yref = csr_apply(A, x);
      double worst = 0.0;
      for (int gi = 0; gi < e.ngroups; ++gi)
        for (int g = 0; g < e.G; ++g) {
          cd acc(0, 0);
          for (int k = 0; k < e.K; ++k)
            acc += e.vals[((size_t)gi * e.K + k) * e.G + g] * xi[e.cols[(size_t)gi * e.K + k]];
          const int rint = gi * e.G + g;
          const int rnat = rows_int.empty() ? rint : rows_int[rint];
          worst = std::max(worst, std::abs(acc - yref[rnat]));
        }
      CHECK(worst < 1e-12, "ell apply mismatch %g (shape %dx%d variant %d)", worst, sh[0], sh[1],
            variant);
    }
  }
  // error paths: bad column index, non-monotone indptr, empty operator, short maps
  {
    Csr A = random_csr(8, 8, 3, rng, false);
    swp::EllHost e;
    std::string err;
    std::vector<int> none;
    if (!A.indices.empty()) {
      Csr B = A;
      B.indices[0] = 99;
      CHECK(swp::ell_pack(e, err, 8, 8, B.indptr.data(), B.indices.data(), B.data.data(), none, none) != 0,
            "out-of-range column accepted");
    }
    Csr Cm = A;
    Cm.indptr[3] = Cm.indptr[4] + 1;
    CHECK(swp::ell_pack(e, err, 8, 8, Cm.indptr.data(), Cm.indices.data(), Cm.data.data(), none, none) != 0,
          "non-monotone indptr accepted");
    CHECK(swp::ell_pack(e, err, 0, 8, A.indptr.data(), A.indices.data(), A.data.data(), none, none) != 0,
          "empty operator accepted");
    std::vector<int> shortmap(3, 0);
    CHECK(swp::ell_pack(e, err, 8, 8, A.indptr.data(), A.indices.data(), A.data.data(), shortmap, none) != 0,
          "short row map accepted");
  }
}

static void test_bsr(std::mt19937& rng) {
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int n : {16, 64, 160}) {
    Csr A = random_csr(n, n, 12, rng, true);
    swp::BsrHost b;
    swp::bsr_pack(b, n, A.indptr.data(), A.indices.data(), A.data.data(), 0.0);
    CHECK(b.KS > 0 && b.KS % 4 == 0, "bsr_pack KS %d", b.KS);
    if (b.KS == 0) continue;
    // the diagonal block's four column groups are the last four k-steps of every row tile
    for (int rt = 0; rt < n / 16; ++rt)
      for (int d = 0; d < 4; ++d)
        CHECK(b.kcol[(size_t)rt * b.KS + b.KS - 4 + d] == 16 * rt + 4 * d, "diagonal block not last (tile %d)", rt);
    std::vector<cd> x(n);
    for (auto& v : x) v = cd(u(rng), u(rng));
    const std::vector<cd> yref = csr_apply(A, x);
    double worst = 0.0;
    for (int rt = 0; rt < n / 16; ++rt)
      for (int i = 0; i < 16; ++i) {
        cd acc(0, 0);
        for (int k = 0; k < b.KS; ++k)
          for (int c = 0; c < 4; ++c)
            acc += b.vals[((size_t)rt * b.KS + k) * 64 + i + 16 * c] *
                   x[b.kcol[(size_t)rt * b.KS + k] + c];
        worst = std::max(worst, std::abs(acc - yref[rt * 16 + i]));
      }
    CHECK(worst < 1e-12, "bsr apply mismatch %g (n %d)", worst, n);
  }
  Csr A = random_csr(24, 24, 3, rng, false);   // n % 16 != 0: must decline, not crash
  swp::BsrHost b;
  swp::bsr_pack(b, 24, A.indptr.data(), A.indices.data(), A.data.data(), 0.0);
  CHECK(b.KS == 0, "bsr_pack accepted n %% 16 != 0");
  Csr S = random_csr(64, 64, 1, rng, false);   // sparse: below the fill threshold
  swp::bsr_pack(b, 64, S.indptr.data(), S.indices.data(), S.data.data(), 0.6);
  CHECK(b.KS == 0, "bsr_pack ignored min_fill");
}

static void test_mt() {
  // stream vs the published MT19937 known answer for seed 5489 (first output 3499211612)
  sw_mt19937* g = sw_mt_create(5489u);
  uint32_t first = 0;
  sw_mt_raw(g, 1, &first);
  CHECK(first == 3499211612u, "MT19937(5489) first output %u", first);
  sw_mt_destroy(g);
  const uint64_t Js[] = {0, 1, 623, 624, 625, 1248, 20000, 1000003};
  for (uint64_t J : Js) {
    sw_mt19937* a = sw_mt_create(123456u);
    sw_mt19937* b = sw_mt_create(123456u);
    uint32_t pre[37];
    sw_mt_raw(a, 37, pre);
    sw_mt_raw(b, 37, pre);
    sw_mt_skip(a, J);
    CHECK(sw_mt_jump(b, J) == 0, "sw_mt_jump failed");
    uint32_t x[700], y[700];
    sw_mt_raw(a, 700, x);
    sw_mt_raw(b, 700, y);
    CHECK(std::memcmp(x, y, sizeof x) == 0, "jump %llu != walk", (unsigned long long)J);
    // state round trip and window
    uint32_t key[624], win[624], win2[624];
    int pos = -1;
    sw_mt_get_state(a, key, &pos);
    sw_mt19937* c = sw_mt_from_state(key, pos);
    sw_mt_window(a, win);
    sw_mt_window(c, win2);
    CHECK(std::memcmp(win, win2, sizeof win) == 0, "window after state round trip");
    uint32_t wj[624];
    CHECK(sw_mt_window_jump(win, 5000, wj) == 0, "window jump failed");
    sw_mt_skip(a, 5000);
    sw_mt_window(a, win2);
    CHECK(std::memcmp(wj, win2, sizeof wj) == 0, "window jump != walk");
    int8_t r8[64], z8[64];
    sw_mt_rademacher(a, 64, r8);
    sw_mt_z4(c, 64, z8);
    for (int i = 0; i < 64; ++i) CHECK(r8[i] == 1 || r8[i] == -1, "rademacher code");
    for (int i = 0; i < 64; ++i) CHECK(z8[i] == 1 || z8[i] == -1 || z8[i] == 2 || z8[i] == -2, "z4 code");
    sw_mt_destroy(a);
    sw_mt_destroy(b);
    sw_mt_destroy(c);
  }
  uint32_t poly[624];
  CHECK(sw_mt_jump_poly(1, poly) == 0 && poly[0] == 2u, "x^1 mod phi must be x");
  CHECK(sw_mt_from_state(nullptr, 0) == nullptr, "null key accepted");
}

// product form of the smoother polynomials (sw_poly.hpp): q(z) = (1 - prod (1 - w_k z)) / z against
// beta prod (1 - u_j z) at points of the spectral region, degrees as the solvers use them (8, 10) and the
// degree-64 polynomial of BASELINE config 2's even-odd smoother; degenerate inputs
static void test_poly(std::mt19937& rng) {
  typedef std::complex<long double> lc;
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int nu : {3, 7, 8, 10, 24, 64}) {
    // weights 1 / theta_k, theta_k Chebyshev-like points of [0.25, 7.6] pushed off the real axis
    std::vector<std::complex<double>> w, u;
    for (int k = 0; k < nu; ++k) {
      const double c = std::cos(M_PI * (k + 0.5) / nu);
      const std::complex<double> th(3.925 + 3.675 * c, 0.3 * (U(rng) - 0.5));
      w.push_back(1.0 / th);
    }
    std::shuffle(w.begin(), w.end(), rng);
    std::complex<double> beta;
    CHECK(swp::product_form(w, u, beta), "product form not found");
    CHECK((int)u.size() == nu - 1, "product form: number of factors");
    double worst = 0.0;
    for (int t = 0; t < 50; ++t) {
      const lc z(0.2 + 7.6 * U(rng), 0.4 * (U(rng) - 0.5));
      lc p = 1.0L;
      for (auto& wk : w) p *= (lc(1.0L) - lc(wk.real(), wk.imag()) * z);
      const lc q = (lc(1.0L) - p) / z;
      lc v(beta.real(), beta.imag());
      for (auto& uj : u) v *= (lc(1.0L) - lc(uj.real(), uj.imag()) * z);
      worst = std::max(worst, (double)(std::abs(v - q) / std::max(std::abs(q), (long double)1e-3L)));
    }
    CHECK(worst < 1e-8, "product form does not reproduce q");
  }
  std::vector<std::complex<double>> w2 = {{0.5, 0.0}, {0.25, 0.0}}, u2;
  std::complex<double> b2;
  CHECK(!swp::product_form(w2, u2, b2) && u2.empty(), "degree below 3 accepted");
  std::vector<std::complex<double>> w3 = {{0.5, 0.0}, {0.0, 0.0}, {0.25, 0.1}};
  CHECK(!swp::product_form(w3, u2, b2) && u2.empty(), "zero weight accepted");
}

int main() {
  std::mt19937 rng(20240);
  test_ell(rng);
  test_bsr(rng);
  test_mt();
  test_poly(rng);
  if (fails) {
    std::printf("host_sanitize: %d check(s) failed\n", fails);
    return 1;
  }
  std::printf("host_sanitize: all checks passed\n");
  return 0;
}
