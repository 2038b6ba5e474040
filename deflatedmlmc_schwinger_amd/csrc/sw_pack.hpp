// sw_pack.hpp -- host-side packing of CSR operators into the engine's device formats (pure C++,
// no HIP: also built with -fsanitize=address,undefined by `make sanitize`).
//   grouped ELL  (k_ell):       G consecutive rows share one sorted list of K column indices;
//                               cols[ngroups][K], vals[ngroups][K][G], padding col 0 / val 0
//   MFMA block-row (k_bsr_mfma): per 16-row tile a sorted list of KS 4-column groups ("k-steps");
//                               kcol[RT][KS] = first column of the group, vals[RT][KS][64] with
//                               lane = (row & 15) + 16 * (col & 3)
// The block structures are those of SURVEY 3.4 (multigrid.py:192-280 products).
#pragma once
#include <algorithm>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace swp {

struct EllHost {
  int K = 0, G = 1, ngroups = 0;
  std::vector<int> cols;
  std::vector<std::complex<double>> vals;
  // optional processing order of the row groups (empty: natural).  For tall operators
  // (prolongators) whose consecutive row groups do NOT share columns -- level 0 rows are in
  // even-odd lattice order, an aggregate's rows are spread over both parity halves -- the groups
  // are visited sorted by their first column, so the waves of a workgroup and the row band of an
  // XCD read the same coarse rows (each fetched into ONE L2, once).
  std::vector<int> order;
};

struct BsrHost {
  int KS = 0;
  std::vector<int> kcol;
  std::vector<std::complex<double>> vals;
};

// rows_int[r] = natural row stored at internal row r (empty: identity);
// colmap[c]   = internal column of natural column c (empty: identity)
inline int ell_pack(EllHost& out, std::string& err, int nrows, int ncols, const int64_t* indptr,
                    const int32_t* indices, const std::complex<double>* data,
                    const std::vector<int>& rows_int, const std::vector<int>& colmap,
                    int forceG = 0, bool sort_by_column = false) {
  char buf[160];
  if (nrows <= 0 || ncols <= 0) {
    err = "build_ell: empty operator";
    return 1;
  }
  if ((!rows_int.empty() && (int)rows_int.size() != nrows) ||
      (!colmap.empty() && (int)colmap.size() != ncols)) {
    err = "build_ell: row/column map of the wrong length";
    return 1;
  }
  for (int r = 0; r < nrows; ++r) {
    if (indptr[r + 1] < indptr[r]) {
      err = "build_ell: indptr not monotone";
      return 1;
    }
    for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q)
      if (indices[q] < 0 || indices[q] >= ncols) {
        std::snprintf(buf, sizeof buf, "build_ell: column index %d out of range [0,%d)", indices[q],
                      ncols);
        err = buf;
        return 1;
      }
  }
  for (int r : rows_int)
    if (r < 0 || r >= nrows) {
      err = "build_ell: row map entry out of range";
      return 1;
    }
  for (int c : colmap)
    if (c < 0 || c >= ncols) {
      err = "build_ell: column map entry out of range";
      return 1;
    }
  auto natrow = [&](int r) { return rows_int.empty() ? r : rows_int[r]; };
  auto icol = [&](int c) { return colmap.empty() ? c : colmap[c]; };
  int bestG = 1;
  int bestK = 0;
  double bestCost = 1e300;
  const int cand[5] = {16, 8, 4, 2, 1};
  std::vector<int> u;
  for (int ci = 0; ci < 5; ++ci) {
    const int G = cand[ci];
    if (forceG && G != forceG) continue;
    if (nrows % G) continue;
    int K = 0;
    for (int g0 = 0; g0 < nrows; g0 += G) {
      u.clear();
      for (int g = 0; g < G; ++g) {
        const int r = natrow(g0 + g);
        for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) u.push_back(icol(indices[q]));
      }
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      K = std::max(K, (int)u.size());
    }
    if (K == 0) K = 1;
    const double cost = K * (1.0 / G + 0.25);
    if (cost < bestCost) {
      bestCost = cost;
      bestG = G;
      bestK = K;
    }
  }
  if (bestK == 0) {
    err = "build_ell: no admissible row grouping";
    return 1;
  }
  const int G = bestG, K = bestK, ng = nrows / G;
  out.G = G;
  out.K = K;
  out.ngroups = ng;
  out.cols.assign((size_t)ng * K, 0);
  out.vals.assign((size_t)ng * K * G, std::complex<double>(0, 0));
  for (int gi = 0; gi < ng; ++gi) {
    u.clear();
    for (int g = 0; g < G; ++g) {
      const int r = natrow(gi * G + g);
      for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) u.push_back(icol(indices[q]));
    }
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    for (size_t k = 0; k < u.size(); ++k) out.cols[(size_t)gi * K + k] = u[k];
    for (int g = 0; g < G; ++g) {
      const int r = natrow(gi * G + g);
      for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) {
        const int c = icol(indices[q]);
        const size_t k = std::lower_bound(u.begin(), u.end(), c) - u.begin();
        out.vals[((size_t)gi * K + k) * G + g] += data[q];
      }
    }
  }
  out.order.clear();
  if (sort_by_column) {
    out.order.resize(ng);
    for (int gi = 0; gi < ng; ++gi) out.order[gi] = gi;
    std::stable_sort(out.order.begin(), out.order.end(), [&](int a, int b) {
      return out.cols[(size_t)a * K] < out.cols[(size_t)b * K];
    });
    bool identity = true;
    for (int gi = 0; gi < ng; ++gi) identity = identity && out.order[gi] == gi;
    if (identity) out.order.clear();
  }
  return 0;
}

// KS == 0 on return: the operator does not qualify (n % 16, empty, or fill below min_fill)
inline void bsr_pack(BsrHost& out, int n, const int64_t* indptr, const int32_t* indices,
                     const std::complex<double>* data, double min_fill) {
  out.KS = 0;
  out.kcol.clear();
  out.vals.clear();
  if (n <= 0 || n % 16) return;
  const int RT = n / 16;
  std::vector<std::vector<int>> groups(RT);
  int KS = 0;
  for (int rt = 0; rt < RT; ++rt) {
    std::vector<int>& g = groups[rt];
    for (int r = rt * 16; r < rt * 16 + 16; ++r)
      for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) g.push_back(indices[q] >> 2);
    // the four column groups of the diagonal block always take part (zeros where absent) ...
    for (int d = 0; d < 4; ++d) g.push_back(rt * 4 + d);
    std::sort(g.begin(), g.end());
    g.erase(std::unique(g.begin(), g.end()), g.end());
    // ... and come LAST, in order: the smoother kernel then finds its own X rows in the operand
    // registers of the final four k-steps (k_bsr_mfma, xreg)
    std::vector<int> off, diag;
    for (int c : g) (c >= rt * 4 && c < rt * 4 + 4 ? diag : off).push_back(c);
    g = off;
    g.insert(g.end(), diag.begin(), diag.end());
    KS = std::max(KS, (int)g.size());
  }
  if (KS == 0) return;
  KS = (KS + 3) & ~3;   // whole four-stage rounds of the kernel's register pipeline
  const double fill = (double)indptr[n] / ((double)RT * KS * 64.0);
  if (fill < min_fill) return;
  out.kcol.assign((size_t)RT * KS, 0);
  out.vals.assign((size_t)RT * KS * 64, std::complex<double>(0, 0));
  std::vector<int> slot_of;   // column group -> k-step of this row tile
  for (int rt = 0; rt < RT; ++rt) {
    const std::vector<int>& g = groups[rt];
    // padding (zero values, a valid column) sits BEFORE the diagonal block, which stays last
    const size_t pad = (size_t)KS - g.size(), nd = g.size() - 4;
    for (size_t k = 0; k < (size_t)KS; ++k) {
      const size_t src = k < nd ? k : (k < nd + pad ? nd : k - pad);
      out.kcol[(size_t)rt * KS + k] = g[src] * 4;
    }
    for (int i = 0; i < 16; ++i) {
      const int r = rt * 16 + i;
      for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) {
        const int c = indices[q];
        size_t k = std::find(g.begin(), g.end(), c >> 2) - g.begin();
        if (k >= nd) k += pad;
        const int lane = i + 16 * (c & 3);
        out.vals[((size_t)rt * KS + k) * 64 + lane] += data[q];
      }
    }
  }
  out.KS = KS;
}

}  // namespace swp
