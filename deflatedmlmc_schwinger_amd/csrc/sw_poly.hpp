// Host-side polynomial algebra of the fixed polynomial smoothers (no device code).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace swp {

// The nu smoother steps x <- x + w_k (b - S x) give x + q(S)(b - S x) with
//   q(z) = (1 - p(z)) / z,   p(z) = prod_k (1 - w_k z)        (degree nu - 1).
// product_form writes q as  q(z) = beta prod_j (1 - u_j z),  beta = q(0) = sum_k w_k,  u_j = 1 / theta_j with
// theta_j the roots of q: Aberth iteration on q evaluated through the product (no monomial coefficients, so
// any degree), started between the zeros 1 / w_k of p; the roots Leja-ordered (largest modulus first, then
// farthest from those taken).  Accepted only if beta prod_j (1 - u_j z) reproduces q at the nu points
// z = 1 / w_k, where q(z) = w_k, to 1e-9 relative; returns false otherwise (and for nu < 3).
inline bool product_form(const std::vector<std::complex<double>>& w_in, std::vector<std::complex<double>>& u_out,
                         std::complex<double>& beta_out) {
  typedef std::complex<long double> lc;
  u_out.clear();
  const int nu = (int)w_in.size();
  if (nu < 3) return false;
  std::vector<lc> w(nu);
  for (int k = 0; k < nu; ++k) {
    w[k] = lc(w_in[k].real(), w_in[k].imag());
    if (std::abs(w[k]) == 0.0L) return false;
  }
  const lc one(1.0L);
  auto qval = [&](lc z, lc* dq) -> lc {
    lc p = one;
    for (int k = 0; k < nu; ++k) p *= (one - w[k] * z);
    lc dp = 0.0L;                      // p'(z) = sum_k (-w_k) prod_{i != k} (1 - w_i z)
    for (int k = 0; k < nu; ++k) {
      lc t = -w[k];
      for (int i = 0; i < nu; ++i)
        if (i != k) t *= (one - w[i] * z);
      dp += t;
    }
    const lc q = (one - p) / z;
    if (dq) *dq = (-dp - q) / z;
    return q;
  };
  const int n = nu - 1;
  std::vector<lc> th(nu);
  for (int k = 0; k < nu; ++k) th[k] = one / w[k];
  std::sort(th.begin(), th.end(), [](const lc& a, const lc& b) { return a.real() < b.real(); });
  std::vector<lc> z(n);
  for (int j = 0; j < n; ++j)
    z[j] = (th[j] + th[j + 1]) * 0.5L + lc(0.0L, 1e-3L * std::abs(th[j + 1] - th[j]) + 1e-6L);
  bool converged = false;
  for (int it = 0; it < 100 && !converged; ++it) {
    long double worst = 0.0L;
    for (int j = 0; j < n; ++j) {
      lc dq;
      const lc q = qval(z[j], &dq);
      if (std::abs(dq) == 0.0L) {
        z[j] += lc(1e-6L, 1e-6L);
        worst = 1.0L;
        continue;
      }
      const lc newton = q / dq;
      lc rep = 0.0L;
      for (int i = 0; i < n; ++i)
        if (i != j) rep += one / (z[j] - z[i]);
      const lc step = newton / (one - newton * rep);
      z[j] -= step;
      worst = std::max(worst, std::abs(step) / std::max(std::abs(z[j]), (long double)1e-30L));
    }
    converged = worst < 1e-17L;
  }
  if (!converged) return false;
  std::vector<lc> ord;
  std::vector<char> used(n, 0);
  for (int t = 0; t < n; ++t) {
    int best = -1;
    long double bv = 0.0L;
    for (int j = 0; j < n; ++j) {
      if (used[j]) continue;
      long double v = 0.0L;
      if (t == 0) v = std::abs(z[j]);
      else
        for (const lc& c : ord) v += std::log(std::max(std::abs(z[j] - c), (long double)1e-300L));
      if (best < 0 || v > bv) {
        best = j;
        bv = v;
      }
    }
    used[best] = 1;
    ord.push_back(z[best]);
  }
  lc beta = 0.0L;
  for (int k = 0; k < nu; ++k) beta += w[k];
  for (int k = 0; k < nu; ++k) {
    const lc zt = one / w[k];
    lc v = beta;
    for (const lc& c : ord) v *= (one - zt / c);
    if (!(std::abs(v - w[k]) <= 1e-9L * std::abs(w[k]))) return false;
  }
  for (const lc& c : ord) {
    const lc u = one / c;
    u_out.emplace_back((double)u.real(), (double)u.imag());
  }
  beta_out = std::complex<double>((double)beta.real(), (double)beta.imag());
  return true;
}

}  // namespace swp
