// sw_kernels.hpp -- CDNA4 (gfx950) kernels of the batched multi-RHS Schwinger engine.
//
// Data layout in HBM: every level vector is a row-major [n][nbp] array of
// complex128 with the right-hand side (probe) index fastest; nbp is a multiple of 64 so one
// wave64 reads/writes one 1-KiB contiguous row segment (16 B per lane) per instruction.
// Level 0 rows are ordered [parity][site-in-parity][spin]  (even-odd lattice layout),
// coarse levels keep the reference's row order.  In every kernel lane == probe, so per-probe
// reductions run down the rows inside a thread and need no cross-lane traffic; cross-wave
// sums go through LDS, cross-workgroup sums through a deterministic two-stage reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace swk {

typedef double2 cplx;

#define SW_WAVE 64
#define SW_BLOCK 256
#define SW_WAVES_PER_BLOCK 4

__device__ __forceinline__ cplx cmake(double a, double b) { cplx r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return cmake(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
// conj(a) * b
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {
  return cmake(fma(a.x, b.x, a.y * b.y), fma(a.x, b.y, -a.y * b.x));
}
// acc += a*b
__device__ __forceinline__ void cfma(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
__device__ __forceinline__ void cfmac(cplx& acc, cplx a, cplx b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(-a.y, b.x, acc.y);
}
// i*a
__device__ __forceinline__ cplx cmuli(cplx a) { return cmake(-a.y, a.x); }
__device__ __forceinline__ double rfma(double a, double b, double c) { return fma(a, b, c); }

// complex64 twins (single-precision preconditioner): same formulae on float2
typedef float2 cplxf;
__device__ __forceinline__ float rfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ cplxf cmake(float a, float b) { cplxf r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ cplxf cadd(cplxf a, cplxf b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplxf csub(cplxf a, cplxf b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplxf cmul(cplxf a, cplxf b) {
  return cmake(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplxf cmulc(cplxf a, cplxf b) {
  return cmake(fmaf(a.x, b.x, a.y * b.y), fmaf(a.x, b.y, -a.y * b.x));
}
__device__ __forceinline__ void cfma(cplxf& acc, cplxf a, cplxf b) {
  acc.x = fmaf(a.x, b.x, acc.x);
  acc.x = fmaf(-a.y, b.y, acc.x);
  acc.y = fmaf(a.x, b.y, acc.y);
  acc.y = fmaf(a.y, b.x, acc.y);
}
__device__ __forceinline__ cplxf cmuli(cplxf a) { return cmake(-a.y, a.x); }
// two complex64 probes per lane: keeps the 16-B-per-lane (1 KiB per wave) accesses of the fp64 layout
// in the HBM-bound complex64 kernels; a [n][nbp] complex64 array is read as [n][nbp/2] of these
struct alignas(16) cplxf2 {
  cplxf a, b;
};
__device__ __forceinline__ cplxf2 cadd(cplxf2 p, cplxf2 q) { return cplxf2{cadd(p.a, q.a), cadd(p.b, q.b)}; }
__device__ __forceinline__ cplxf2 csub(cplxf2 p, cplxf2 q) { return cplxf2{csub(p.a, q.a), csub(p.b, q.b)}; }
__device__ __forceinline__ cplxf2 cmuli(cplxf2 p) { return cplxf2{cmuli(p.a), cmuli(p.b)}; }
// scalar (link / weight) times pair
__device__ __forceinline__ cplxf2 cmul(cplxf u, cplxf2 p) { return cplxf2{cmul(u, p.a), cmul(u, p.b)}; }
__device__ __forceinline__ cplxf2 cmulc(cplxf u, cplxf2 p) { return cplxf2{cmulc(u, p.a), cmulc(u, p.b)}; }
__device__ __forceinline__ void cfma(cplxf2& acc, cplxf u, cplxf2 p) {
  cfma(acc.a, u, p.a);
  cfma(acc.b, u, p.b);
}
template <class C> struct real_of;
template <> struct real_of<cplx> { typedef double type; };
template <> struct real_of<cplxf> { typedef float type; };
template <> struct real_of<cplxf2> { typedef float type; };
// the scalar complex type that multiplies a vector element (gauge links, weights)
template <class C> struct scalar_of { typedef C type; };
template <> struct scalar_of<cplxf2> { typedef cplxf type; };
// alpha u + beta v, real alpha and beta
__device__ __forceinline__ cplx clin(double al, cplx u, double be, cplx v) {
  return cmake(fma(al, u.x, be * v.x), fma(al, u.y, be * v.y));
}
__device__ __forceinline__ cplxf clin(float al, cplxf u, float be, cplxf v) {
  return cmake(fmaf(al, u.x, be * v.x), fmaf(al, u.y, be * v.y));
}
__device__ __forceinline__ cplxf2 clin(float al, cplxf2 u, float be, cplxf2 v) {
  return cplxf2{clin(al, u.a, be, v.a), clin(al, u.b, be, v.b)};
}
// q - d c + di acc  (residual of the Schur complement: b' - (D x - acc / D))
__device__ __forceinline__ cplx cschur(cplx q, double d, cplx c, double di, cplx acc) {
  return cmake(q.x - d * c.x + di * acc.x, q.y - d * c.y + di * acc.y);
}
__device__ __forceinline__ cplxf cschur(cplxf q, float d, cplxf c, float di, cplxf acc) {
  return cmake(q.x - d * c.x + di * acc.x, q.y - d * c.y + di * acc.y);
}
__device__ __forceinline__ cplxf2 cschur(cplxf2 q, float d, cplxf2 c, float di, cplxf2 acc) {
  return cplxf2{cschur(q.a, d, c.a, di, acc.a), cschur(q.b, d, c.b, di, acc.b)};
}
__device__ __forceinline__ cplx widen(cplx v) { return v; }
__device__ __forceinline__ cplx widen(cplxf v) { return cmake((double)v.x, (double)v.y); }
template <class C> __device__ __forceinline__ C narrow(cplx v);
template <> __device__ __forceinline__ cplx narrow<cplx>(cplx v) { return v; }
template <> __device__ __forceinline__ cplxf narrow<cplxf>(cplx v) { return cmake((float)v.x, (float)v.y); }
template <class C> __device__ __forceinline__ C czero() {
  C r;
  r.x = 0;
  r.y = 0;
  return r;
}
template <> __device__ __forceinline__ cplxf2 czero<cplxf2>() {
  return cplxf2{cmake(0.f, 0.f), cmake(0.f, 0.f)};
}

// Blocks are dealt round-robin over the 8 XCDs (MI355X_MICROARCH, Workgroup dispatch); this
// bijective remap hands each XCD one contiguous range of logical blocks so that neighbouring
// lattice rows are served from the same 4-MiB L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  return (nblk % 8 == 0) ? (b % 8) * (nblk / 8) + (b / 8) : b;
}

// ------------------------------------------------------------------------------------------
// Level-0 operator: U(1) Wilson-Schwinger stencil  A = S + m  (replaces the CSR
// SpMV of multigrid.py:552-557 / matrix.py:21-29).  One wave = SPW lattice sites x 64 probes.
//   (S psi)(n) = 4 psi(n) - [ (1-s1) U1(n) psi(n+x) + (1+s1) U1*(n-x) psi(n-x)
//                           + (1-s2) U2(n) psi(n+y) + (1+s2) U2*(n-y) psi(n-y) ]
// MODE 0: Y = A X      MODE 1: Y = B - A X      MODE 2: Y = X + w (B - A X)  (one fused
// Richardson/polynomial-smoother step: 3 vector passes, no inner products)
// ------------------------------------------------------------------------------------------
template <class C>
struct StencilArgsT {
  int L;          // lattice extent (even)
  int Vh;         // L*L/2
  typename real_of<C>::type diag;    // 4 + mass
  const typename scalar_of<C>::type* U1;    // [L*L] site index y*L+x
  const typename scalar_of<C>::type* U2;
  int nbp;             // row length in elements of C
  int tile_w;          // x-extent of the lattice tiles the blocks walk (divides L)
  typename scalar_of<C>::type w;   // MODE 2 relaxation weight
  typename scalar_of<C>::type w2 = {};   // k_schur_step MODE 4: weight of the fused update
  int nt_store;        // non-temporal output stores
  // even-odd kernels only: work on the lattice rows row0 .. row0 + nrows - 1 (mod L) instead of all of
  // them (nrows = 0): the time-skewed order of the smoother's steps on lattices beyond the Infinity Cache
  int row0 = 0, nrows = 0;
};
typedef StencilArgsT<cplx> StencilArgs;

__device__ __forceinline__ size_t eo_row(int x, int y, int L, int Vh) {
  // row of spin 0 of site (x,y) in the even-odd layout; spin 1 is the next row
  int par = (x + y) & 1;
  int sh = (y * L + x) >> 1;
  return ((size_t)par * Vh + sh) * 2;
}

// CI: storage type of X (complex64 when X is a direction made by the single-precision
// preconditioner; widened on load -- the arithmetic is fp64 either way)
template <int MODE, int SPW, class CI = cplx, class CO = cplx>
__global__ __launch_bounds__(SW_BLOCK) void k_stencil(const CI* __restrict__ X,
                                                      const cplx* __restrict__ B,
                                                      CO* __restrict__ Y, StencilArgs a,
                                                      int blocks_per_chunk) {
  const int nblk = gridDim.x;
  const int bb = xcd_remap(blockIdx.x, nblk);
  const int chunk = bb / blocks_per_chunk;
  const int sg = bb % blocks_per_chunk;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int L = a.L;
  const int nbp = a.nbp;
  const size_t col = (size_t)chunk * 64 + lane;
  const int site0 = __builtin_amdgcn_readfirstlane((sg * SW_WAVES_PER_BLOCK + wave) * SPW);
  if (site0 >= L * L) return;
  // tile-major walk: tiles of tile_w x L sites, row by row inside a tile, so that the three
  // lattice-row segments a block's neighbours live in (3 * tile_w * 2 KiB per 64-probe chunk)
  // stay resident in the XCD's 4-MiB L2 on large lattices
  const int tw = a.tile_w;
  const int tile = site0 / (tw * L);
  const int rem = site0 - tile * (tw * L);
  const int y = rem / tw;
  const int x0 = tile * tw + (rem - y * tw);
  const int yp = (y + 1 == L) ? 0 : y + 1;
  const int ym = (y == 0) ? L - 1 : y - 1;
  const CI* Xc = X + col;
#define SW_LDX(ROW) widen(Xc[(ROW) * nbp])
  // one wave = SPW consecutive x-sites of one lattice row x 64 probes; the x-neighbours slide
  // through registers (left, centre, right), so a site costs 3 site-loads instead of 5
  const int xl = (x0 == 0) ? L - 1 : x0 - 1;
  size_t r_l = eo_row(xl, y, L, a.Vh);
  size_t r_c = eo_row(x0, y, L, a.Vh);
  cplx l0 = SW_LDX(r_l), l1 = SW_LDX(r_l + 1);
  cplx c0 = SW_LDX(r_c), c1 = SW_LDX(r_c + 1);
  cplx u1m = a.U1[y * L + xl];
#pragma unroll
  for (int s = 0; s < SPW; ++s) {
    const int x = x0 + s;  // SPW divides tile_w, so the run never leaves the tile row
    const int xp = (x + 1 == L) ? 0 : x + 1;
    const size_t r_xp = eo_row(xp, y, L, a.Vh);
    const size_t r_yp = eo_row(x, yp, L, a.Vh);
    const size_t r_ym = eo_row(x, ym, L, a.Vh);
    const cplx a0 = SW_LDX(r_xp), a1 = SW_LDX(r_xp + 1);
    const cplx d0 = SW_LDX(r_yp), d1 = SW_LDX(r_yp + 1);
    const cplx e0 = SW_LDX(r_ym), e1 = SW_LDX(r_ym + 1);
    const int n = y * L + x;
    const cplx u1 = a.U1[n];
    const cplx u2 = a.U2[n];
    const cplx u2m = a.U2[ym * L + x];
    // +x: (1-s1) -> [t,-t], t = psi0 - psi1
    const cplx tx = cmul(u1, csub(a0, a1));
    // -x: (1+s1) -> [t, t], t = psi0 + psi1, link conj(U1(n-x))
    const cplx txm = cmulc(u1m, cadd(l0, l1));
    // +y: (1-s2) -> [t, -i t], t = psi0 + i psi1
    const cplx ty = cmul(u2, cadd(d0, cmuli(d1)));
    // -y: (1+s2) -> [t, i t], t = psi0 - i psi1, link conj(U2(n-y))
    const cplx tym = cmulc(u2m, csub(e0, cmuli(e1)));
    cplx h0 = cadd(cadd(tx, txm), cadd(ty, tym));
    cplx h1 = cadd(csub(txm, tx), cmuli(csub(tym, ty)));
    cplx o0 = cmake(fma(a.diag, c0.x, -h0.x), fma(a.diag, c0.y, -h0.y));
    cplx o1 = cmake(fma(a.diag, c1.x, -h1.x), fma(a.diag, c1.y, -h1.y));
    if (MODE == 1) {
      const cplx q0 = B[r_c * nbp + col], q1 = B[(r_c + 1) * nbp + col];
      o0 = csub(q0, o0);
      o1 = csub(q1, o1);
    }
    if (MODE == 2) {
      const cplx q0 = B[r_c * nbp + col], q1 = B[(r_c + 1) * nbp + col];
      cplx t0 = c0, t1 = c1;
      cfma(t0, a.w, csub(q0, o0));
      cfma(t1, a.w, csub(q1, o1));
      o0 = t0;
      o1 = t1;
    }
    if (sizeof(CO) == sizeof(cplx) && a.nt_store) {
      // streaming output: keep it from displacing the neighbour rows held in L2
      double* y0 = (double*)&Y[r_c * nbp + col];
      double* y1 = (double*)&Y[(r_c + 1) * nbp + col];
      __builtin_nontemporal_store(o0.x, y0);
      __builtin_nontemporal_store(o0.y, y0 + 1);
      __builtin_nontemporal_store(o1.x, y1);
      __builtin_nontemporal_store(o1.y, y1 + 1);
    } else {
      Y[r_c * nbp + col] = narrow<CO>(o0);
      Y[(r_c + 1) * nbp + col] = narrow<CO>(o1);
    }
    // slide the window
    l0 = c0; l1 = c1;
    c0 = a0; c1 = a1;
    r_c = r_xp;
    u1m = u1;
  }
#undef SW_LDX
}

// ------------------------------------------------------------------------------------------
// a 2-spinor at one lattice site (both spins x 64 probes per wave) and its load
// ------------------------------------------------------------------------------------------
template <class C>
struct SiteT {
  C s0, s1;
};
typedef SiteT<cplx> Site2;

template <class C>
__device__ __forceinline__ SiteT<C> ld_site(const C* __restrict__ base, int x, int y, int L, int Vh,
                                            int nbp) {
  const size_t r = eo_row(x, y, L, Vh);
  SiteT<C> v;
  v.s0 = base[r * nbp];
  v.s1 = base[(r + 1) * nbp];
  return v;
}

// ------------------------------------------------------------------------------------------
// Even-odd (Schur-complement) smoothing of the stencil level.  With the rows in even-odd order,
//   A = [ D  -H_eo ; -H_oe  D ],  D = 4 + m,   S = D - H_eo H_oe / D   (even sites only)
// and A^-1 b follows from  b'_e = b_e + H_eo b_o / D,  S x_e = b'_e,  x_o = (b_o + H_oe x_e) / D.
// The post-smoother iterates x_e <- x_e + w_k (b'_e - S x_e) on HALF vectors and then sets x_o
// exactly: errors on the odd sites and the operator's upper spectrum are gone after one step,
// and S is four times better conditioned than A.  The two hops of S are ONE kernel: the
// back-tracking paths vanish identically ((1 -+ s_mu)(1 +- s_mu) = 0), leaving the 8 even sites
// at lattice distance 2 (16 row loads + x_e, b'_e: 20 per output site, L2-served), and the HBM
// traffic of a step is 3 half-vector passes = 201 MB instead of 403 MB (128^2, 256 probes).
// ------------------------------------------------------------------------------------------
// contribution of one hop to a 2-spinor: dir 0:+x (1-s1) u psi, 1:-x (1+s1) conj(u) psi,
// 2:+y (1-s2) u psi, 3:-y (1+s2) conj(u) psi   (u = the link the hop runs along)
template <int DIR, class C>
__device__ __forceinline__ void hop_acc(SiteT<C>& acc, typename scalar_of<C>::type u, SiteT<C> psi) {
  if (DIR == 0) {
    const C t = cmul(u, csub(psi.s0, psi.s1));
    acc.s0 = cadd(acc.s0, t);
    acc.s1 = csub(acc.s1, t);
  } else if (DIR == 1) {
    const C t = cmulc(u, cadd(psi.s0, psi.s1));
    acc.s0 = cadd(acc.s0, t);
    acc.s1 = cadd(acc.s1, t);
  } else if (DIR == 2) {
    const C t = cmul(u, cadd(psi.s0, cmuli(psi.s1)));
    acc.s0 = cadd(acc.s0, t);
    acc.s1 = csub(acc.s1, cmuli(t));
  } else {
    const C t = cmulc(u, csub(psi.s0, cmuli(psi.s1)));
    acc.s0 = cadd(acc.s0, t);
    acc.s1 = cadd(acc.s1, cmuli(t));
  }
}

__device__ __forceinline__ int wrap_p(int v, int L) { return (v + 1 == L) ? 0 : v + 1; }
__device__ __forceinline__ int wrap_m(int v, int L) { return (v == 0) ? L - 1 : v - 1; }

// site (x, y) of parity q for work item `it` in [0, V/2): the items walk x-tiles of tw lattice
// columns (tw/2 sites of one parity per row), row by row inside a tile, so that the five lattice
// rows a Schur step touches stay in the XCD's L2 on large lattices (as k_stencil's tile walk)
__device__ __forceinline__ void eo_site(int it, int q, int L, int tw, int row0, int rows, int& x, int& y) {
  const int ht = tw >> 1;
  const int tile = it / (ht * rows);
  const int rem = it - tile * (ht * rows);
  const int yl = rem / ht;
  y = row0 + yl;
  if (y >= L) y -= L;
  x = 2 * (tile * ht + (rem - yl * ht)) + ((q + y) & 1);
}

// out(n) = alpha * Uv(n) + beta * (H src)(n)  on the sites n of parity Q; src on the other parity.
//   Q = 0, Uv = src = B, alpha = 1, beta = 1/D :  b'_e = b_e + H_eo b_o / D
//   Q = 1, Uv = B, src = X, alpha = beta = 1/D:  x_o  = (b_o + H_oe x_e) / D
template <int Q, class C>
__global__ __launch_bounds__(SW_BLOCK) void k_eo_hop(const C* __restrict__ Uv,
                                                     const C* __restrict__ src,
                                                     C* __restrict__ out, StencilArgsT<C> a,
                                                     typename real_of<C>::type alpha,
                                                     typename real_of<C>::type beta,
                                                     int blocks_per_chunk) {
  const int bb = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = bb / blocks_per_chunk;
  const int sg = bb % blocks_per_chunk;
  const int lane = threadIdx.x & 63;
  const int L = a.L, Vh = a.Vh, nbp = a.nbp;
  const size_t col = (size_t)chunk * 64 + lane;
  const int sh = __builtin_amdgcn_readfirstlane(sg * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  const int rows = a.nrows > 0 ? a.nrows : L;
  if (sh >= rows * (L >> 1)) return;
  int x, y;
  eo_site(sh, Q, L, a.tile_w, a.row0, rows, x, y);
  const int xp = wrap_p(x, L), xm = wrap_m(x, L), yp = wrap_p(y, L), ym = wrap_m(y, L);
  const C* S = src + col;
  SiteT<C> acc;
  acc.s0 = czero<C>();
  acc.s1 = acc.s0;
  hop_acc<0>(acc, a.U1[y * L + x], ld_site(S, xp, y, L, Vh, nbp));
  hop_acc<1>(acc, a.U1[y * L + xm], ld_site(S, xm, y, L, Vh, nbp));
  hop_acc<2>(acc, a.U2[y * L + x], ld_site(S, x, yp, L, Vh, nbp));
  hop_acc<3>(acc, a.U2[ym * L + x], ld_site(S, x, ym, L, Vh, nbp));
  const size_t r = eo_row(x, y, L, Vh);
  const C u0 = Uv[r * nbp + col], u1 = Uv[(r + 1) * nbp + col];
  out[r * nbp + col] = clin(alpha, u0, beta, acc.s0);
  out[(r + 1) * nbp + col] = clin(alpha, u1, beta, acc.s1);
}

// Y_e = X_e + w (Bp_e - S X_e),  S = D - H_eo H_oe / D : both hops in one kernel (even sites only)
// MODE 2: that smoother step;  MODE 0: Y_e = S X_e;  MODE 1: Y_e = Bp_e - S X_e  (operator and residual
// of the even-odd reduced system the outer Krylov solver works on, fgmres_eo);
// MODE 3: Y_e = X_e - w S X_e  (one factor of the smoother polynomial in product form: Bp is not read);
// MODE 4: Y_e = Bp_e + w2 (X_e - w S X_e)  (the last factor, fused with the update of the iterate Bp; may run
// in place, Y = Bp: a site reads Bp only at itself)
template <class C, int MODE = 2>
__global__ __launch_bounds__(SW_BLOCK) void k_schur_step(const C* __restrict__ X, const C* Bp, C* Y,
                                                         StencilArgsT<C> a, int blocks_per_chunk) {
  typedef SiteT<C> Site2;
  typedef typename real_of<C>::type real;
  const int bb = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = bb / blocks_per_chunk;
  const int sg = bb % blocks_per_chunk;
  const int lane = threadIdx.x & 63;
  const int L = a.L, Vh = a.Vh, nbp = a.nbp;
  const size_t col = (size_t)chunk * 64 + lane;
  const int sh = __builtin_amdgcn_readfirstlane(sg * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  const int rows = a.nrows > 0 ? a.nrows : L;
  if (sh >= rows * (L >> 1)) return;
  int x, y;
  eo_site(sh, 0, L, a.tile_w, a.row0, rows, x, y);
  const int xp = wrap_p(x, L), xm = wrap_m(x, L), yp = wrap_p(y, L), ym = wrap_m(y, L);
  const int xpp = wrap_p(xp, L), xmm = wrap_m(xm, L), ypp = wrap_p(yp, L), ymm = wrap_m(ym, L);
  const C* Xc = X + col;
  // the eight even sites at distance 2
  const Site2 e20 = ld_site(Xc, xpp, y, L, Vh, nbp), em20 = ld_site(Xc, xmm, y, L, Vh, nbp);
  const Site2 e02 = ld_site(Xc, x, ypp, L, Vh, nbp), e0m2 = ld_site(Xc, x, ymm, L, Vh, nbp);
  const Site2 ePP = ld_site(Xc, xp, yp, L, Vh, nbp), ePM = ld_site(Xc, xp, ym, L, Vh, nbp);
  const Site2 eMP = ld_site(Xc, xm, yp, L, Vh, nbp), eMM = ld_site(Xc, xm, ym, L, Vh, nbp);
  const typename scalar_of<C>::type* U1 = a.U1;
  const typename scalar_of<C>::type* U2 = a.U2;
#define SW_U1(xx, yy) U1[(yy) * L + (xx)]
#define SW_U2(xx, yy) U2[(yy) * L + (xx)]
  Site2 z;
  z.s0 = czero<C>();
  z.s1 = z.s0;
  // t(o) = sum over the hops into the odd neighbour o that do not come from n; then the hop o -> n
  Site2 t = z;                                          // o = n + x
  hop_acc<0>(t, SW_U1(xp, y), e20);
  hop_acc<2>(t, SW_U2(xp, y), ePP);
  hop_acc<3>(t, SW_U2(xp, ym), ePM);
  Site2 acc = z;
  hop_acc<0>(acc, SW_U1(x, y), t);
  t = z;                                                // o = n - x
  hop_acc<1>(t, SW_U1(xmm, y), em20);
  hop_acc<2>(t, SW_U2(xm, y), eMP);
  hop_acc<3>(t, SW_U2(xm, ym), eMM);
  hop_acc<1>(acc, SW_U1(xm, y), t);
  t = z;                                                // o = n + y
  hop_acc<2>(t, SW_U2(x, yp), e02);
  hop_acc<0>(t, SW_U1(x, yp), ePP);
  hop_acc<1>(t, SW_U1(xm, yp), eMP);
  hop_acc<2>(acc, SW_U2(x, y), t);
  t = z;                                                // o = n - y
  hop_acc<3>(t, SW_U2(x, ymm), e0m2);
  hop_acc<0>(t, SW_U1(x, ym), ePM);
  hop_acc<1>(t, SW_U1(xm, ym), eMM);
  hop_acc<3>(acc, SW_U2(x, ym), t);
#undef SW_U1
#undef SW_U2
  const size_t r = eo_row(x, y, L, Vh);
  const real d = a.diag, di = (real)1 / a.diag;
  const C c0 = Xc[r * nbp], c1 = Xc[(r + 1) * nbp];
  C q0 = czero<C>(), q1 = q0;
  if (MODE != 0 && MODE != 3) {
    q0 = Bp[r * nbp + col];
    q1 = Bp[(r + 1) * nbp + col];
  }
  // residual of S: b' - (D x - acc / D)
  const C z0 = czero<C>();
  const C r0 = cschur(MODE == 4 ? z0 : q0, d, c0, di, acc.s0);
  const C r1 = cschur(MODE == 4 ? z0 : q1, d, c1, di, acc.s1);
  C o0, o1;
  if (MODE == 0) {          // S x = -(0 - S x)
    o0 = csub(czero<C>(), r0);
    o1 = csub(czero<C>(), r1);
  } else if (MODE == 1) {
    o0 = r0;
    o1 = r1;
  } else if (MODE == 4) {
    C v0 = c0, v1 = c1;
    cfma(v0, a.w, r0);
    cfma(v1, a.w, r1);
    o0 = q0;
    o1 = q1;
    cfma(o0, a.w2, v0);
    cfma(o1, a.w2, v1);
  } else {
    o0 = c0;
    o1 = c1;
    cfma(o0, a.w, r0);
    cfma(o1, a.w, r1);
  }
  Y[r * nbp + col] = o0;
  Y[(r + 1) * nbp + col] = o1;
}

// ------------------------------------------------------------------------------------------
// One site of the Schur step as a function of its operands -- the 8 even neighbours at distance 2, the
// site's own x, b' -- and of a LINK SOURCE, written in exactly the order of k_schur_step above (bit-identical
// results): shared by the LDS-tiled persistent kernel below.  (A persistent REGISTER-pipelined form of
// k_schur_step -- two workgroups per CU walking the items, the next item's 20 rows loading into a second
// register set, 212-228 VGPRs -- was built on it, bit-identical, and measured 51 us against 35 per launch:
// with two waves per SIMD only 160 KiB of loads are in flight per CU; profiles/r04_ab_sessions.txt, r04d.)
// ------------------------------------------------------------------------------------------
template <class C>
struct SchurLoads {
  SiteT<C> e20, em20, e02, e0m2, ePP, ePM, eMP, eMM;
  C c0, c1, q0, q1;
};

// link sources: the 16 link values of one output site, m = 4 * (hop into o) + (position inside it)
//   o = n + x: U1(xp,y) U2(xp,y) U2(xp,ym) U1(x,y)    o = n - x: U1(xmm,y) U2(xm,y) U2(xm,ym) U1(xm,y)
//   o = n + y: U2(x,yp) U1(x,yp) U1(xm,yp) U2(x,y)    o = n - y: U2(x,ymm) U1(x,ym) U1(xm,ym) U2(x,ym)
// packed tables of (dx + 2, dy + 2) in 2 bits each and of the direction bit (1: U2), indexed by m
#define SW_LK_DX 0x9a9a54bfu   // dx + 2 for m = 0..15: 3 3 3 2 0 1 1 1 2 2 1 2 2 2 1 2
#define SW_LK_DY 0x54bf9a9au   // dy + 2 for m = 0..15: 2 2 1 2 2 2 1 2 3 3 3 2 0 1 1 1
#define SW_LK_U2 0x9966u       // U2 for m = 1,2,5,6,8,11,12,15

// links from a 1-KiB LDS row this wave gathered by LDS-DMA: value (q, m) at slot q * 16 + m
struct LdsLinks {
  const cplx* row;   // + q * 16 already applied
  template <int M>
  __device__ __forceinline__ cplx get() const { return row[M]; }
};

template <class C, int MODE, class LK>
__device__ __forceinline__ void schur_site_compute(const SchurLoads<C>& ld, C* __restrict__ Yc, int x, int y,
                                                   const StencilArgsT<C>& a, const LK& lk) {
  typedef SiteT<C> Site2;
  typedef typename real_of<C>::type real;
  const int L = a.L, Vh = a.Vh, nbp = a.nbp;
  Site2 z;
  z.s0 = czero<C>();
  z.s1 = z.s0;
  Site2 t = z;                                          // o = n + x
  hop_acc<0>(t, lk.template get<0>(), ld.e20);
  hop_acc<2>(t, lk.template get<1>(), ld.ePP);
  hop_acc<3>(t, lk.template get<2>(), ld.ePM);
  Site2 acc = z;
  hop_acc<0>(acc, lk.template get<3>(), t);
  t = z;                                                // o = n - x
  hop_acc<1>(t, lk.template get<4>(), ld.em20);
  hop_acc<2>(t, lk.template get<5>(), ld.eMP);
  hop_acc<3>(t, lk.template get<6>(), ld.eMM);
  hop_acc<1>(acc, lk.template get<7>(), t);
  t = z;                                                // o = n + y
  hop_acc<2>(t, lk.template get<8>(), ld.e02);
  hop_acc<0>(t, lk.template get<9>(), ld.ePP);
  hop_acc<1>(t, lk.template get<10>(), ld.eMP);
  hop_acc<2>(acc, lk.template get<11>(), t);
  t = z;                                                // o = n - y
  hop_acc<3>(t, lk.template get<12>(), ld.e0m2);
  hop_acc<0>(t, lk.template get<13>(), ld.ePM);
  hop_acc<1>(t, lk.template get<14>(), ld.eMM);
  hop_acc<3>(acc, lk.template get<15>(), t);
  const size_t r = eo_row(x, y, L, Vh);
  const real d = a.diag, di = (real)1 / a.diag;
  const C c0 = ld.c0, c1 = ld.c1, q0 = ld.q0, q1 = ld.q1;
  const C z0 = czero<C>();
  const C r0 = cschur(MODE == 4 ? z0 : q0, d, c0, di, acc.s0);
  const C r1 = cschur(MODE == 4 ? z0 : q1, d, c1, di, acc.s1);
  C o0, o1;
  if (MODE == 0) {
    o0 = csub(czero<C>(), r0);
    o1 = csub(czero<C>(), r1);
  } else if (MODE == 1) {
    o0 = r0;
    o1 = r1;
  } else if (MODE == 4) {
    C v0 = c0, v1 = c1;
    cfma(v0, a.w, r0);
    cfma(v1, a.w, r1);
    o0 = q0;
    o1 = q1;
    cfma(o0, a.w2, v0);
    cfma(o1, a.w2, v1);
  } else {
    o0 = c0;
    o1 = c1;
    cfma(o0, a.w, r0);
    cfma(o1, a.w, r1);
  }
  Yc[r * nbp] = o0;
  Yc[(r + 1) * nbp] = o1;
}

// ------------------------------------------------------------------------------------------
// The same step from an LDS-staged halo tile, double-buffered by LDS-DMA (engine option "eo_tile").
// k_schur_step asks the L2 for 20 rows of 1 KiB per output site and chunk, 18 of them re-reads of rows
// its neighbours also ask for: 694 MB through the L2 -> CU path per launch at 128^2 x 256 probes, the
// ~70-77 GB/s per CU that path delivers for gathers, for 134-202 MB of unique bytes.  Here the even sites
// are tiled in the ROTATED coordinates u = (x + y) / 2, v = (x - y) / 2 -- in which the eight even sites at
// lattice distance 2 are exactly the 3 x 3 neighbourhood: (x,y) = ((u + v) mod L, (u - v) mod L) maps
// u in [0, L), v in [0, L/2) one-to-one onto the even sites, and because the formula is periodic a halo
// needs no wrap logic -- and a workgroup stages a 4 x 4 tile plus its halo ring (36 sites x 2 spins x 1 KiB
// = 72 KiB) in LDS: 2.25 site loads per output site instead of 9.  One persistent workgroup per CU walks its
// share of the (chunk, tile) jobs; while it computes tile k from one LDS buffer, the LDS-DMA loads of tile
// k + 1 (global_load_lds_dwordx4: no VGPR destination, queued by the same waves before they start computing)
// land in the other.  Synchronisation per tile: a COUNTED s_waitcnt vmcnt(stores of the previous tile) -- the
// DMA loads of this tile were issued before those stores and VMEM operations retire in order, so they have
// landed, while the stores stay in flight -- and one raw s_barrier (no __syncthreads: its fence would drain
// the counter to zero).  XCD q's workgroups sweep the contiguous job band q, side by side, so the halo rows
// neighbouring tiles share meet in that XCD's L2.  Arithmetic: schur_site_compute, bit-identical results.
// MODE 0 / 3 (no b' operand).  L % 8 == 0.
// ------------------------------------------------------------------------------------------
#define SW_TILE_ROWS 72   // (4 + 2) x (4 + 2) sites x 2 spins

__device__ __forceinline__ void uv_site(int uu, int vv, int L, int& x, int& y) {
  x = uu + vv;
  y = uu - vv;
  if (x < 0) x += L;
  if (x >= L) x -= L;
  if (y < 0) y += L;
  if (y >= L) y -= L;
}

template <int WAVES>
__device__ __forceinline__ void schur_tile_issue(const cplx* __restrict__ X, unsigned lds_base, int u0, int v0,
                                                 int wave, int lane, size_t col, int L, int Vh, int nbp) {
  constexpr int RPW = SW_TILE_ROWS / WAVES;
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    const int rho = wave * RPW + t;
    const int st = rho >> 1, spin = rho & 1;
    const int i = st / 6 - 1, j = st % 6 - 1;
    int x, y;
    uv_site(u0 + i, v0 + j, L, x, y);
    const cplx* g = X + (eo_row(x, y, L, Vh) + spin) * nbp + col;
    // global_load_lds_dwordx4 by hand (M0 = wave-uniform LDS byte address, written in the same statement): the
    // builtin would work, but hipcc then puts s_waitcnt vmcnt(0) in front of the first ds_read that follows
    // (its LDS-DMA alias bookkeeping), which serialises the prefetch with the compute it is meant to overlap
    const unsigned dst = lds_base + (unsigned)rho * 1024u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst));
  }
}

__device__ __forceinline__ Site2 lds_site(const cplx* lds_buf, int i, int j, int lane) {
  const int st = (i + 1) * 6 + (j + 1);
  Site2 v;
  v.s0 = lds_buf[(st * 2) * 64 + lane];
  v.s1 = lds_buf[(st * 2 + 1) * 64 + lane];
  return v;
}

// this wave's link gather for the tile at (u0, v0): lane q * 16 + m fetches link m of the wave's q-th output
// site -- ONE LDS-DMA instruction with per-lane source addresses -- into the 1-KiB row at lds_dst
template <int WAVES>
__device__ __forceinline__ void schur_tile_links(const cplx* __restrict__ U1, const cplx* __restrict__ U2,
                                                 unsigned lds_dst, int u0, int v0, int wave, int lane, int L) {
  constexpr int SPW = 16 / WAVES;
  const int q = lane >> 4, m = lane & 15;
  const int sidx = wave * SPW + (q < SPW ? q : 0);
  int x, y;
  uv_site(u0 + (sidx >> 2), v0 + (sidx & 3), L, x, y);
  int xx = x + (int)((SW_LK_DX >> (2 * m)) & 3u) - 2;
  int yy = y + (int)((SW_LK_DY >> (2 * m)) & 3u) - 2;
  if (xx < 0) xx += L;
  if (xx >= L) xx -= L;
  if (yy < 0) yy += L;
  if (yy >= L) yy -= L;
  const cplx* g = (((SW_LK_U2 >> m) & 1u) ? U2 : U1) + (size_t)yy * L + xx;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_dst));
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k_schur_tile(const cplx* __restrict__ X, cplx* __restrict__ Y,
                                                              StencilArgs a, int tiles_v, int ntiles, int njobs) {
  static_assert(MODE == 0 || MODE == 3, "tile kernel: modes without a b' operand");
  // per buffer: 72 vector rows + one link row per wave
  constexpr int BUF_ROWS = SW_TILE_ROWS + WAVES;
  __shared__ cplx lds[2 * BUF_ROWS * 64];
  constexpr int SPW = 16 / WAVES;            // output sites per wave and tile
  constexpr int VM_PER_TILE = 2 * SPW;       // this wave's stores per tile (issued after its DMA loads)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wgx = blockIdx.x >> 3, nwgx = gridDim.x >> 3;
  const int band0 = (int)(((long long)njobs * xcd) >> 3);
  const int band1 = (int)(((long long)njobs * (xcd + 1)) >> 3);
  const int L = a.L, Vh = a.Vh, nbp = a.nbp;
  int job = band0 + wgx;
  if (job >= band1) return;
  int cur = 0;
  // LDS byte address of the tile buffers (the value of the address-space-3 pointer)
  const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) cplx*)lds);
  {
    const int chunk = job / ntiles, tile = job - chunk * ntiles;
    const int u0 = (tile / tiles_v) * 4, v0 = (tile % tiles_v) * 4;
    schur_tile_issue<WAVES>(X, lds0, u0, v0, wave, lane, (size_t)chunk * 64 + lane, L, Vh, nbp);
    schur_tile_links<WAVES>(a.U1, a.U2, lds0 + (unsigned)(SW_TILE_ROWS + wave) * 1024u, u0, v0, wave, lane, L);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (;;) {
    // tile `job` has landed in buffer `cur` (this wave's rows: counted wait below / prologue above); the barrier
    // makes every wave's rows visible and says everybody is done reading the other buffer
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");      // (compiler fence: no LDS read of this tile above the barrier)
    const int chunk = job / ntiles, tile = job - chunk * ntiles;
    const int u0 = (tile / tiles_v) * 4, v0 = (tile % tiles_v) * 4;
    const size_t col = (size_t)chunk * 64 + lane;
    const int nxt = job + nwgx;
    const bool more = nxt < band1;
    if (more && !(a.row0 & 1)) {      // (a.row0: diagnostic bits of option eo_tile_dbg -- 1: no prefetch, 2: no compute)
      const int c2 = nxt / ntiles, t2 = nxt - c2 * ntiles;
      const int u2 = (t2 / tiles_v) * 4, v2 = (t2 % tiles_v) * 4;
      const unsigned b2 = lds0 + (unsigned)(cur ^ 1) * (BUF_ROWS * 1024u);
      schur_tile_issue<WAVES>(X, b2, u2, v2, wave, lane, (size_t)c2 * 64 + lane, L, Vh, nbp);
      schur_tile_links<WAVES>(a.U1, a.U2, b2 + (unsigned)(SW_TILE_ROWS + wave) * 1024u, u2, v2, wave, lane, L);
    }
    const cplx* buf = lds + (size_t)cur * BUF_ROWS * 64;
    if (!(a.row0 & 2))
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
      const int sidx = wave * SPW + q;
      const int i = sidx >> 2, j = sidx & 3;
      SchurLoads<cplx> ld;
      ld.e20 = lds_site(buf, i + 1, j + 1, lane);
      ld.em20 = lds_site(buf, i - 1, j - 1, lane);
      ld.e02 = lds_site(buf, i + 1, j - 1, lane);
      ld.e0m2 = lds_site(buf, i - 1, j + 1, lane);
      ld.ePP = lds_site(buf, i + 1, j, lane);
      ld.ePM = lds_site(buf, i, j + 1, lane);
      ld.eMP = lds_site(buf, i, j - 1, lane);
      ld.eMM = lds_site(buf, i - 1, j, lane);
      const Site2 own = lds_site(buf, i, j, lane);
      ld.c0 = own.s0;
      ld.c1 = own.s1;
      ld.q0 = cmake(0.0, 0.0);
      ld.q1 = ld.q0;
      int x, y;
      uv_site(u0 + i, v0 + j, L, x, y);
      LdsLinks lk;
      lk.row = buf + (size_t)(SW_TILE_ROWS + wave) * 64 + q * 16;
      schur_site_compute<cplx, MODE>(ld, Y + col, x, y, a, lk);
    }
    if (!more) return;
    // this wave's DMA loads of the next tile were issued BEFORE its 2 SPW stores above: all but the last
    // 2 SPW VMEM operations retired = they have landed; its LDS reads of `cur` are complete as well
    if (a.row0 & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM_PER_TILE) : "memory");
    job = nxt;
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// Grouped-ELL operator: coarse operators A_l, prolongators P_l, restrictors R_l = P_l^H, the
// dense coarsest inverse and the MLMC rhs maps.  G consecutive rows share one list of K column
// indices (the dense-block structure of SURVEY 3.4); one wave = one row group x 64 probes, each
// X row is loaded once and used G times with wave-uniform (scalar) coefficients.
//   cols [ngroups][K]   vals [ngroups][K][G]   (padding: col 0, val 0)
// MODE 0: Y = A X      MODE 1: Y = B - A X      MODE 2: Y = B + A X
// MODE 3: Y = X + w (B - A X)   (square operators only)
// ------------------------------------------------------------------------------------------
template <int G, int MODE, class C = cplx>
__global__ __launch_bounds__(SW_BLOCK) void k_ell(const int* __restrict__ cols,
                                                  const C* __restrict__ vals, int K,
                                                  int ngroups, const int* __restrict__ order,
                                                  const C* __restrict__ X,
                                                  const C* __restrict__ B,
                                                  C* __restrict__ Y, int nbp, C w) {
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int bx = xcd_remap(blockIdx.x, gridDim.x);   // contiguous row band per XCD (L2 reuse)
  const int slot = __builtin_amdgcn_readfirstlane(bx * SW_WAVES_PER_BLOCK + wave);
  if (slot >= ngroups) return;
  const int grp = order ? __builtin_amdgcn_readfirstlane(order[slot]) : slot;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int* c = cols + (size_t)grp * K;
  const C* v = vals + (size_t)grp * K * G;
  C acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = czero<C>();
  const C* Xc = X + col;
  // eight X rows requested at a time, then their G x 8 multiply-adds (coefficients are scalar loads): left to
  // itself the compiler keeps two row loads in flight and a K = 32 group pays sixteen round trips (the restrictor
  // of the lattice level: 23 us per launch for 75 MB).  Same order of the sums per row.
  constexpr int UC = (G <= 8) ? 8 : 4;
  int k = 0;
  for (; k + UC <= K; k += UC) {
    C xs[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) xs[u] = Xc[(size_t)c[k + u] * nbp];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UC; ++u) {
#pragma unroll
      for (int g = 0; g < G; ++g) cfma(acc[g], v[(size_t)(k + u) * G + g], xs[u]);
    }
  }
  for (; k < K; ++k) {
    const C x = Xc[(size_t)c[k] * nbp];
#pragma unroll
    for (int g = 0; g < G; ++g) cfma(acc[g], v[(size_t)k * G + g], x);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t row = (size_t)grp * G + g;
    C o = acc[g];
    if (MODE == 1) o = csub(B[row * nbp + col], o);
    if (MODE == 2) o = cadd(B[row * nbp + col], o);
    if (MODE == 3) {
      C t = X[row * nbp + col];
      cfma(t, w, csub(B[row * nbp + col], o));
      o = t;
    }
    Y[row * nbp + col] = o;
  }
}

// ------------------------------------------------------------------------------------------
// Block-row operator on the fp64 matrix cores: the dense coarsest-level inverse
// (multigrid.py:413-416) and block-structured coarse operators.
//   Y[n][nbp] (op)= A[n][n] * X[n][nbp]   (complex128, A given per 16-row tile as a list of
//   4-column groups: "k-steps")
// v_mfma_f64_16x16x4_f64:  D(16x16) += A(16x4) B(4x16),  lane l holds A[l&15][l>>4],
// B[l>>4][l&15] and D[(l>>4)+4r][l&15], r<4 (cdna_hip_programming.md section 3).
// X and Y are used as REAL [n][2*nbp] matrices (re/im interleaved along the row), so one
// 16-column MFMA tile covers 8 probes.  Two accumulators per tile: D1 = Re(A) X'', D2 = Im(A) X'';
//   Y''[i][2j] = D1[i][2j] - D2[i][2j+1],   Y''[i][2j+1] = D1[i][2j+1] + D2[i][2j]
// i.e. one neighbour-lane exchange in the epilogue.  A is pre-packed at upload time so every
// k-step of a tile is one coalesced 1-KiB read:  Ap[(rt*KS + ks)*64 + lane] =
// A[rt*16 + (lane&15)][kcol[rt*KS+ks] + (lane>>4)].
// One wave = one 16-row tile x NT MFMA column tiles (8 probes each): NT = 8 keeps 16
// accumulators (one wave per SIMD), NT = 4 halves the registers (two waves per SIMD: the pure
// issue rate of v_mfma_f64_16x16x4 is 36 TF/s at one wave/SIMD and 47 TF/s at two,
// tools/mfma_f64_rate.hip).
// MODE 0: Y = A X   MODE 1: Y = B - A X   MODE 3: Y = X + w (B - A X)
// ------------------------------------------------------------------------------------------
typedef double sw_double4 __attribute__((ext_vector_type(4)));
// waves per SIMD the register allocator must leave room for (NT = 4 with 4 stages lands on 128 + 64
// registers = 2 waves without it)
#ifndef SW_BSR_WAVES_NT4
#define SW_BSR_WAVES_NT4 3
#endif
#define SW_BSR_MIN_WAVES(NT_, STG_) (((NT_) == 4 && (STG_) <= 4) ? SW_BSR_WAVES_NT4 : 1)

// epilogue of the block-row kernels: re/im exchange between neighbour lanes, mode arithmetic, store
// xown (MODE 3, optional): the wave's own X values xown[r][t] = X''[16 ot + (lane>>4) + 4r][c0 + 16t + c],
// left in the operand registers by the diagonal block's four k-steps when these come last
template <int MODE, int NT, bool NTIO>
__device__ __forceinline__ void bsr_store(const sw_double4 (&re)[NT], const sw_double4 (&im)[NT], int rt,
                                          int c0, int lane, const int* __restrict__ tmap,
                                          const double* __restrict__ Xr,
                                          const double* __restrict__ Br, double* __restrict__ Yr,
                                          int ld, cplx w, const double (*xown)[NT] = nullptr) {
  const int c = lane & 15;
  const bool odd = (c & 1) != 0;
  const int ot = tmap ? __builtin_amdgcn_readfirstlane(tmap[rt]) : rt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // tmap (optional): row tile -> tile of the OUTPUT vector it belongs to (operators that act on
    // a subset of a level's sites, e.g. the even-odd Schur operators of a coarse level)
    const size_t row = (size_t)ot * 16 + (lane >> 4) + 4 * r;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const double s = __shfl_xor(im[t][r], 1);
      double y = odd ? re[t][r] + s : re[t][r] - s;
      const size_t off = row * ld + c0 + t * 16 + c;
      // B is read once, by exactly this lane: streamed past L2 (nt) when NTIO, so that the operator's
      // values, which every probe chunk of this XCD's row band re-reads, stay resident there
      // (hoisting these loads above the MFMA loop was measured 5 % SLOWER: register pressure)
      if (MODE == 1) y = (NTIO ? __builtin_nontemporal_load(&Br[off]) : Br[off]) - y;
      if (MODE == 3) {
        const double tt = (NTIO ? __builtin_nontemporal_load(&Br[off]) : Br[off]) - y;
        const double tp = __shfl_xor(tt, 1);
        y = (xown ? xown[r][t] : Xr[off]) + w.x * tt + (odd ? w.y * tp : -w.y * tp);
      }
      if (NTIO) __builtin_nontemporal_store(y, &Yr[off]);
      else Yr[off] = y;
    }
  }
}

template <int MODE, int NT, bool NTIO = false, int STG = 2>
__global__ __launch_bounds__(SW_BLOCK, SW_BSR_MIN_WAVES(NT, STG)) void k_bsr_mfma(const cplx* __restrict__ Ap,
                                                       const int* __restrict__ kcol, int KS,
                                                       int RT, const double* __restrict__ Xr,
                                                       const double* __restrict__ Br,
                                                       double* __restrict__ Yr, int ld, int nbp,
                                                       cplx w, int map, int msub,
                                                       const int* __restrict__ tmap, int xreg) {
  // the 4 waves of a workgroup take 4 consecutive row tiles and the SAME 64-probe chunk, so
  // the X rows they share (all of them for a dense operator, the common neighbours for a
  // block stencil) are served once from L2 and then from the CU's L1
  const int lane = threadIdx.x & 63;
  // XCD-aware: consecutive block ids are dealt round-robin over the 8 XCDs; remapping hands each
  // XCD one contiguous band of row tiles, so the X rows its tiles share (lattice neighbours of a
  // block stencil) are fetched into that XCD's L2 once instead of into all eight
  // 1-D grid of RB x NC blocks (RB row blocks of 4 tiles, NC probe chunks); `map` orders them:
  //   0: chunk-major, the row blocks of one chunk dealt to the XCDs in contiguous bands
  //   1: XCD band of row blocks, the NC chunks of a row block adjacent in time (A tile read once)
  //   2: as 1, in sub-bands of `msub` row blocks: chunk loop over a sub-band, then the next one
  // (four row tiles x one chunk per workgroup; 2 x 2 and 1 x 4 -- chunk-mates sharing each A tile in
  // L1 -- were measured identical: the A stream is not what the kernel waits for)
  const int RB = (RT + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const int NC = (gridDim.y > 1) ? gridDim.y : gridDim.x / RB;
  int bx, cy;
  if (gridDim.y > 1) {            // map 0 is launched as a 2-D grid (RB, NC)
    cy = blockIdx.y;
    bx = xcd_remap(blockIdx.x, RB);
  } else if (map == 0 || map == 3 || (RB & 7)) {
    cy = blockIdx.x / RB;
    bx = xcd_remap(blockIdx.x - cy * RB, RB);
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, RBx = RB >> 3;
    if (map == 1 || (RBx % msub)) {
      cy = j % NC;
      bx = xcd * RBx + j / NC;
    } else {
      const int per = NC * msub;
      const int sidx = j / per, rem = j - sidx * per;
      cy = rem / msub;
      bx = xcd * RBx + sidx * msub + (rem - cy * msub);
    }
  }
  const int rt = __builtin_amdgcn_readfirstlane(bx * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (rt >= RT) return;
  const int c0 = cy * (16 * NT);                    // first real column of this chunk
  const cplx* a = Ap + (size_t)rt * KS * 64 + lane;
  const int* kc = kcol + (size_t)rt * KS;           // wave-uniform -> scalar loads
  const double* b = Xr + (size_t)(lane >> 4) * ld + c0 + (lane & 15);
  sw_double4 re[NT], im[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    re[t] = sw_double4{0.0, 0.0, 0.0, 0.0};
    im[t] = re[t];
  }
  // STG register stages (KS % STG == 0): the loads of k-step ks+STG are issued right behind the
  // MFMAs of k-step ks and have STG-1 k-steps of matrix work to land (SQ counters of the 2-stage
  // version on the level-1 operator: waves parked at s_waitcnt 31 % of their cycles, matrix pipe
  // busy 36 % -- one k-step of MFMAs, 0.4 us for the wave pair, does not cover an L2/fabric round
  // trip); the sched_barriers keep hipcc from sinking the loads below the next MFMA block.
#define SW_BSR_LOAD(M_, X_, KSI)                               \
  {                                                            \
    M_ = a[(size_t)(KSI) * 64];                                \
    const double* bk_ = b + (size_t)kc[(KSI)] * ld;            \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) X_[t] = bk_[t * 16]; \
  }
#define SW_BSR_MFMA(M_, X_)                                                        \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                 \
    re[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(M_.x, X_[t], re[t], 0, 0, 0);     \
    im[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(M_.y, X_[t], im[t], 0, 0, 0);     \
  }
  cplx mm[STG];
  double xx[STG][NT];
#pragma unroll
  for (int s = 0; s < STG; ++s) SW_BSR_LOAD(mm[s], xx[s], s);
  for (int ks = 0; ks < KS; ks += STG) {
#pragma unroll
    for (int s = 0; s < STG; ++s) {
      const int kn = (ks + s + STG < KS) ? ks + s + STG : ks + s;   // tail: harmless re-load
      __builtin_amdgcn_sched_barrier(0);
      SW_BSR_MFMA(mm[s], xx[s]);
      __builtin_amdgcn_sched_barrier(0);
      SW_BSR_LOAD(mm[s], xx[s], kn);
    }
  }
#undef SW_BSR_LOAD
#undef SW_BSR_MFMA
  // xreg: the operator's diagonal block is its last four k-steps (the packers put it there), so after
  // the tail re-loads stage s holds the X rows 16 ot + 4 s + (lane >> 4) -- exactly the rows the
  // smoother update reads, in the accumulator's lane layout: no second trip to memory for x
  if (MODE == 3 && STG == 4 && xreg) bsr_store<MODE, NT, NTIO>(re, im, rt, c0, lane, tmap, Xr, Br, Yr, ld, w, xx);
  else bsr_store<MODE, NT, NTIO>(re, im, rt, c0, lane, tmap, Xr, Br, Yr, ld, w);
}

// ------------------------------------------------------------------------------------------
// The block-row operator with THREE real matrix products per complex one ("3M"):
//   T1 = Ar Xr,  T2 = Ai Xi,  T3 = (Ar + Ai)(Xr + Xi);   Re Y = T1 - T2,   Im Y = T3 - T1 - T2
// -- 3 v_mfma_f64_16x16x4 per (k-step, 16 probes) instead of the 4 of k_bsr_mfma (2 per 8 probes): a
// quarter of the matrix-core time of the dense inverses, which are MFMA-bound.  The sums Ar + Ai and
// Xr + Xi cost one VALU add per loaded value.  X and Y are used as COMPLEX [n][nbp] arrays: a lane
// holds one complex probe value of a 4-row k-step (lane l: row l >> 4 of the step, probe l & 15 of the
// tile: 16 lanes x 16 B = one 256-B segment per row), the accumulators come out as
// D[(l >> 4) + 4 r][l & 15], so a lane owns whole complex results and the epilogue needs no lane
// exchange.  Norm-wise the result is as accurate as the four-product form (error bound
// c eps (|Ar| + |Ai|)(|Xr| + |Xi|)); component-wise Im Y may lose relative accuracy where it cancels,
// which no consumer here depends on (operator applications are compared norm-wise, 1e-13).
// A is packed as for k_bsr_mfma (same kcol, same lane order).  One wave = one 16-row tile x NT tiles of
// 16 probes.  MODE as k_bsr_mfma; xreg: the wave's own X rows from the operand registers of the diagonal
// block's four k-steps (packed last), in exactly the accumulators' lane layout.
// ------------------------------------------------------------------------------------------
#define SW_BSR3_MIN_WAVES(NT_, STG_) (((NT_) <= 2 && (STG_) <= 4) ? 2 : 1)

template <int MODE, int NT, bool NTIO = false, int STG = 4>
__global__ __launch_bounds__(SW_BLOCK, SW_BSR3_MIN_WAVES(NT, STG)) void k_bsr_mfma3(
    const cplx* __restrict__ Ap, const int* __restrict__ kcol, int KS, int RT, const cplx* __restrict__ X,
    const cplx* __restrict__ B, cplx* __restrict__ Y, int nbp, cplx w, int map, int msub,
    const int* __restrict__ tmap, int xreg) {
  const int lane = threadIdx.x & 63;
  // block -> (row block, probe chunk) exactly as k_bsr_mfma (1-D grid of RB x NC blocks)
  const int RB = (RT + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const int NC = gridDim.x / RB;
  int bx, cy;
  if (map == 0 || map == 3 || (RB & 7)) {
    cy = blockIdx.x / RB;
    bx = xcd_remap(blockIdx.x - cy * RB, RB);
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, RBx = RB >> 3;
    if (map == 1 || (RBx % msub)) {
      cy = j % NC;
      bx = xcd * RBx + j / NC;
    } else {
      const int per = NC * msub;
      const int sidx = j / per, rem = j - sidx * per;
      cy = rem / msub;
      bx = xcd * RBx + sidx * msub + (rem - cy * msub);
    }
  }
  const int rt = __builtin_amdgcn_readfirstlane(bx * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (rt >= RT) return;
  const int c0 = cy * (16 * NT);                    // first probe of this chunk
  const cplx* a = Ap + (size_t)rt * KS * 64 + lane;
  const int* kc = kcol + (size_t)rt * KS;           // wave-uniform -> scalar loads
  const cplx* b = X + (size_t)(lane >> 4) * nbp + c0 + (lane & 15);
  sw_double4 t1[NT], t2[NT], t3[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    t1[t] = sw_double4{0.0, 0.0, 0.0, 0.0};
    t2[t] = t1[t];
    t3[t] = t1[t];
  }
#define SW_B3_LOAD(M_, X_, KSI)                                 \
  {                                                             \
    M_ = a[(size_t)(KSI) * 64];                                 \
    const cplx* bk_ = b + (size_t)kc[(KSI)] * nbp;              \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) X_[t] = bk_[t * 16]; \
  }
#define SW_B3_MFMA(M_, X_)                                                                  \
  {                                                                                         \
    const double ms_ = M_.x + M_.y;                                                         \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                        \
      t1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(M_.x, X_[t].x, t1[t], 0, 0, 0);          \
      t2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(M_.y, X_[t].y, t2[t], 0, 0, 0);          \
      t3[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ms_, X_[t].x + X_[t].y, t3[t], 0, 0, 0); \
    }                                                                                       \
  }
  cplx mm[STG];
  cplx xx[STG][NT];
#pragma unroll
  for (int s = 0; s < STG; ++s) SW_B3_LOAD(mm[s], xx[s], (s < KS ? s : 0));
  for (int ks = 0; ks < KS; ks += STG) {
#pragma unroll
    for (int s = 0; s < STG; ++s) {
      if (ks + s < KS) {                                              // KS need not divide by STG
        const int kn = (ks + s + STG < KS) ? ks + s + STG : ks + s;   // tail: harmless re-load
        __builtin_amdgcn_sched_barrier(0);
        SW_B3_MFMA(mm[s], xx[s]);
        __builtin_amdgcn_sched_barrier(0);
        SW_B3_LOAD(mm[s], xx[s], kn);
      }
    }
  }
#undef SW_B3_LOAD
#undef SW_B3_MFMA
  const int ot = tmap ? __builtin_amdgcn_readfirstlane(tmap[rt]) : rt;
  // after the tail re-loads stage s holds the k-step KS - STG + s (KS % STG == 0): with the diagonal block
  // last and STG == 4 these are the X rows 16 ot + 4 s + (lane >> 4) -- row r of this lane's results
  const bool own = MODE == 3 && STG == 4 && xreg && (KS % 4 == 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t row = (size_t)ot * 16 + (lane >> 4) + 4 * r;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const size_t off = row * nbp + c0 + t * 16 + (lane & 15);
      cplx y = cmake(t1[t][r] - t2[t][r], t3[t][r] - t1[t][r] - t2[t][r]);
      if (MODE == 1 || MODE == 3) {
        cplx q;
        if (NTIO) {
          const double* bp_ = (const double*)&B[off];
          q = cmake(__builtin_nontemporal_load(bp_), __builtin_nontemporal_load(bp_ + 1));
        } else {
          q = B[off];
        }
        y = csub(q, y);
      }
      if (MODE == 3) {
        cplx o;
        if constexpr (STG == 4) o = own ? xx[r][t] : X[off];
        else o = X[off];
        cfma(o, w, y);
        y = o;
      }
      if (NTIO) {
        double* yp_ = (double*)&Y[off];
        __builtin_nontemporal_store(y.x, yp_);
        __builtin_nontemporal_store(y.y, yp_ + 1);
      } else {
        Y[off] = y;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// DENSE operator (the coarsest inverse, the dense Schur inverse of a directly solved level, the direct
// inverse of a small level) times a batch, three-product form, operands shared through LDS (engine option
// "dense_lds").  k_bsr_mfma3 on a dense operator loads 2 KiB per wave and k-step (its A fragment and its X
// fragment) for 3 MFMAs = 192 matrix-pipe cycles: at the full matrix rate that is 102 GB/s per CU, more than
// the L2 -> CU path delivers for gathers (~70 GB/s per CU; 2.1 GB through it per launch of the 2048^2 inverse
// on 256 probes).  Here a workgroup of 2 RTW waves computes a (16 RTW)-row x 32-probe block (RTW row tiles x 2
// probe tiles, one pair per wave): per k-step it needs RTW KiB of A and 2 KiB of X for all its waves -- 1 KiB
// (RTW = 2) or 0.75 KiB (RTW = 4) per wave and k-step -- fetched once and shared through LDS.
// Staging: stages of KB (4 or 8) k-steps; every wave owns PPW of a stage's 1-KiB pieces (A: one contiguous piece per
// (row tile, k-step) of the block-row packing as it stands; X: two row pairs x 32 probes per k-step, the rows
// through kcol -- the same for every row tile of a dense operator, `uniform`), loads them with ordinary
// global_load_dwordx4 into a register ring SW_DL_DEPTH stages deep (the compiler counts vmcnt itself), writes
// stage st + 1 into the other LDS buffer while stage st is being multiplied, one barrier per stage.
// (First form, r04i: the same tiling fed by a six-slot ring of LDS-DMA fills, no registers involved --
// bit-identical and no faster than k_bsr_mfma3, 151 against 146 us: the LDS-DMA path of a CU sustains ~25 GB/s,
// MI355X_MICROARCH.md "ldsdma-fill", below the 37 GB/s this tiling needs at the full matrix rate.)
// MODE 0 (Y = A X).  RT % RTW == 0, KS % (4 KB) == 0, nbp % 32 == 0.
// ------------------------------------------------------------------------------------------
#define SW_DL_DEPTH 4

template <int RTW, int KB>
__global__ __launch_bounds__(128 * RTW) void k_dense_mfma3_lds(const cplx* __restrict__ Ap,
                                                              const int* __restrict__ kcol, int KS, int RT,
                                                              const cplx* __restrict__ X, cplx* __restrict__ Y,
                                                              int nbp, const int* __restrict__ tmap) {
  constexpr int WAVES = 2 * RTW;
  constexpr int APIECES = RTW * KB, PIECES = APIECES + 2 * KB, PPW = PIECES / WAVES;
  static_assert(PIECES % WAVES == 0, "pieces must divide over the waves");
  __shared__ cplx lds[2][PIECES * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // block -> (row block of RTW tiles, pair of probe tiles): the NC pairs of a row block adjacent on ONE XCD
  const int RB = RT / RTW, NC = nbp >> 5;
  int rb, pc;
  if ((RB & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, RBx = RB >> 3;
    rb = xcd * RBx + j / NC;
    pc = j % NC;
  } else {
    rb = blockIdx.x / NC;
    pc = blockIdx.x % NC;
  }
  const int rtl = wave >> 1, ptl = wave & 1;
  const int rt0 = rb * RTW;
  const int c0 = pc * 32;
  const int nstages = KS / KB;
  // this wave's pieces of a stage: p = wave + WAVES t.  p < APIECES: A piece (row tile p / KB, k-step p % KB);
  // else X piece q = p - APIECES (k-step q >> 1, row pair q & 1; lane -> row lane >> 5, probe lane & 31).
  // APIECES is a multiple of WAVES, so whether piece t is an A or an X piece is known at compile time.
  // (Everything per piece is a separately named scalar: arrays indexed by t ended up in scratch memory.)
  static_assert(APIECES % WAVES == 0 && PPW >= 3 && PPW <= 6, "piece layout");
#define SW_DL_SRC(T)                                                                                     \
  ((WAVES * (T) < APIECES)                                                                               \
       ? Ap + ((size_t)(rt0 + (wave + WAVES * (T)) / KB) * KS + ((wave + WAVES * (T)) % KB)) * 64 + lane \
       : X + (size_t)(2 * ((wave + WAVES * (T) - APIECES) & 1) + (lane >> 5)) * nbp + c0 + (lane & 31))
  const cplx* const src0 = SW_DL_SRC(0);
  const cplx* const src1 = SW_DL_SRC(1);
  const cplx* const src2 = SW_DL_SRC(2);
  const cplx* const src3 = (PPW > 3) ? SW_DL_SRC(3) : src2;
  const cplx* const src4 = (PPW > 4) ? SW_DL_SRC(4) : src2;
  const cplx* const src5 = (PPW > 5) ? SW_DL_SRC(5) : src2;
#undef SW_DL_SRC
  // k-step (inside a stage) of an X piece
#define SW_DL_XK(T) ((wave + WAVES * (T) - APIECES) >> 1)
  // (stage indices beyond the last are clamped: harmless re-loads, no branches in the loop)
#define SW_DL_LD1(DST, SRC, T, KS0)                                                                \
  if (WAVES * (T) < APIECES) DST = SRC[(size_t)(KS0) * 64];                                        \
  else DST = SRC[(size_t)kcol[(KS0) + SW_DL_XK(T)] * nbp];
#define SW_DL_LOAD(RING, ST)                                                                       \
  {                                                                                                \
    const int ks0_ = min((ST), nstages - 1) * KB;                                                  \
    SW_DL_LD1(RING##_0, src0, 0, ks0_)                                                             \
    SW_DL_LD1(RING##_1, src1, 1, ks0_)                                                             \
    SW_DL_LD1(RING##_2, src2, 2, ks0_)                                                             \
    if (PPW > 3) { SW_DL_LD1(RING##_3, src3, 3, ks0_) }                                            \
    if (PPW > 4) { SW_DL_LD1(RING##_4, src4, 4, ks0_) }                                            \
    if (PPW > 5) { SW_DL_LD1(RING##_5, src5, 5, ks0_) }                                            \
  }
#define SW_DL_WRITE(RING, BUF)                                                                     \
  {                                                                                                \
    lds[BUF][(wave + WAVES * 0) * 64 + lane] = RING##_0;                                           \
    lds[BUF][(wave + WAVES * 1) * 64 + lane] = RING##_1;                                           \
    lds[BUF][(wave + WAVES * 2) * 64 + lane] = RING##_2;                                           \
    if (PPW > 3) lds[BUF][(wave + WAVES * 3) * 64 + lane] = RING##_3;                              \
    if (PPW > 4) lds[BUF][(wave + WAVES * 4) * 64 + lane] = RING##_4;                              \
    if (PPW > 5) lds[BUF][(wave + WAVES * 5) * 64 + lane] = RING##_5;                              \
  }
#define SW_DL_DECL(RING) \
  cplx RING##_0, RING##_1, RING##_2, RING##_3 = cmake(0.0, 0.0), RING##_4 = RING##_3, RING##_5 = RING##_3;
  SW_DL_DECL(ring0)
  SW_DL_DECL(ring1)
  SW_DL_DECL(ring2)
  SW_DL_DECL(ring3)
#undef SW_DL_DECL
  static_assert(SW_DL_DEPTH == 4, "the ring is written out for four stages");
  // one stage: stage ST + 1 (held in RING) into the other LDS buffer -- everybody left it at the barrier that
  // ended stage ST - 1; after the last stage this writes a re-loaded stage nobody reads --, RING refilled with
  // stage ST + 5, then the KB k-steps of stage ST from buffer BUF.  (The fences: hipcc otherwise sinks the
  // refill loads down to their use four stages later and the ring collapses to one stage in flight.)
#define SW_DL_STEP(RING, BUF, ST)                                                                  \
  {                                                                                                \
    SW_DL_WRITE(RING, (BUF) ^ 1);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SW_DL_LOAD(RING, (ST) + 1 + SW_DL_DEPTH);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    const cplx* sl_ = &lds[BUF][0];                                                                \
    _Pragma("unroll") for (int k = 0; k < KB; ++k) {                                               \
      const cplx m_ = sl_[a_off + k * 64];                                                         \
      const cplx xv_ = sl_[x_off + k * 128];                                                       \
      t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(m_.x, xv_.x, t1, 0, 0, 0);                         \
      t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(m_.y, xv_.y, t2, 0, 0, 0);                         \
      t3 = __builtin_amdgcn_mfma_f64_16x16x4f64(m_.x + m_.y, xv_.x + xv_.y, t3, 0, 0, 0);          \
    }                                                                                              \
    __syncthreads();                                                                               \
  }
  // prologue: four stages of loads in flight, stage 0 into LDS buffer 0, its ring slot refilled with stage 4
  SW_DL_LOAD(ring0, 0);
  SW_DL_LOAD(ring1, 1);
  SW_DL_LOAD(ring2, 2);
  SW_DL_LOAD(ring3, 3);
  SW_DL_WRITE(ring0, 0);
  __builtin_amdgcn_sched_barrier(0);
  SW_DL_LOAD(ring0, 4);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  sw_double4 t1 = {0.0, 0.0, 0.0, 0.0}, t2 = t1, t3 = t1;
  // fragment offsets inside a stage buffer: A piece (rtl KB + k) -> [lane]; X pieces (APIECES + 2 k + (row >> 1)) ->
  // [(row & 1) 32 + probe], row = lane >> 4, probe = ptl 16 + (lane & 15)
  const int a_off = (rtl * KB) * 64 + lane;
  const int x_off = (APIECES + (lane >> 5)) * 64 + ((lane >> 4) & 1) * 32 + ptl * 16 + (lane & 15);
  // nstages % 4 == 0 (the launcher checks KS % (4 KB) == 0); stage st + 1 sits in ring (st + 1) % 4
  for (int st = 0; st < nstages; st += 4) {
    SW_DL_STEP(ring1, 0, st);
    SW_DL_STEP(ring2, 1, st + 1);
    SW_DL_STEP(ring3, 0, st + 2);
    SW_DL_STEP(ring0, 1, st + 3);
  }
#undef SW_DL_STEP
#undef SW_DL_LOAD
#undef SW_DL_LD1
#undef SW_DL_XK
#undef SW_DL_WRITE
  const int rt = rt0 + rtl;
  const int ot = tmap ? __builtin_amdgcn_readfirstlane(tmap[rt]) : rt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t row = (size_t)ot * 16 + (lane >> 4) + 4 * r;
    Y[row * nbp + c0 + ptl * 16 + (lane & 15)] = cmake(t1[r] - t2[r], t3[r] - t1[r] - t2[r]);
  }
}

// ------------------------------------------------------------------------------------------
// Single-precision twin of the block-row kernel (f32 preconditioner): v_mfma_f32_16x16x4_f32 runs
// at twice the fp64 matrix rate on gfx950 (157 vs 78.6 TFLOP/s) and every stream is half as wide.
// X and Y are complex64 [n][nbp]; a lane holds ONE complex probe value of a 4-row k-step
// (lane l: row l>>4 of the step, probe l&15 of the tile: 16 lanes x 8 B = one 128-B segment per row),
// so the real and imaginary parts are separate B operands and no lane exchange is needed:
//   Re Y += Re(A) Re(X) + (-Im A) Im(X),   Im Y += Re(A) Im(X) + Im(A) Re(X)     (4 MFMAs per tile)
// C/D of the f32 form: lane l holds rows 4 (l>>4) + r, r < 4, of column l&15 (cdna_hip_programming.md).
// A is packed exactly as for the fp64 kernel (same kcol, same lane order), converted to float2.
// One wave = one 16-row tile x NT tiles of 16 probes.  MODE as k_bsr_mfma.
// ------------------------------------------------------------------------------------------
typedef float sw_float4 __attribute__((ext_vector_type(4)));

template <int MODE, int NT, int STG>
__global__ __launch_bounds__(SW_BLOCK) void k_bsr_mfma_f32(const cplxf* __restrict__ Ap,
                                                           const int* __restrict__ kcol, int KS,
                                                           int RT, const cplxf* __restrict__ X,
                                                           const cplxf* __restrict__ B,
                                                           cplxf* __restrict__ Y, int nbp, cplxf w,
                                                           int map, const int* __restrict__ tmap) {
  const int lane = threadIdx.x & 63;
  // 1-D grid of RB x NC blocks, ordered as k_bsr_mfma's map 0 (chunk-major, XCD bands of row blocks)
  // or map 1 (XCD band of row blocks, the chunks of a row block adjacent in time)
  const int RB = (RT + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const int NC = gridDim.x / RB;
  int bx, cy;
  if (map == 0 || (RB & 7)) {
    cy = blockIdx.x / RB;
    bx = xcd_remap(blockIdx.x - cy * RB, RB);
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, RBx = RB >> 3;
    cy = j % NC;
    bx = xcd * RBx + j / NC;
  }
  const int rt = __builtin_amdgcn_readfirstlane(bx * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (rt >= RT) return;
  const int c0 = cy * (16 * NT);                    // first probe of this chunk
  const cplxf* a = Ap + (size_t)rt * KS * 64 + lane;
  const int* kc = kcol + (size_t)rt * KS;           // wave-uniform -> scalar loads
  const cplxf* b = X + (size_t)(lane >> 4) * nbp + c0 + (lane & 15);
  sw_float4 re[NT], im[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    re[t] = sw_float4{0.f, 0.f, 0.f, 0.f};
    im[t] = re[t];
  }
#define SW_BSRF_LOAD(M_, X_, KSI)                               \
  {                                                             \
    M_ = a[(size_t)(KSI) * 64];                                 \
    const cplxf* bk_ = b + (size_t)kc[(KSI)] * nbp;             \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) X_[t] = bk_[t * 16]; \
  }
  // the two accumulations into re[t] (and im[t]) are 2 NT MFMAs apart: beyond the 40-cycle
  // dependent latency of the instruction for every NT >= 1
#define SW_BSRF_MFMA(M_, X_)                                                              \
  {                                                                                       \
    const float nay_ = -M_.y;                                                             \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                      \
      re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(M_.x, X_[t].x, re[t], 0, 0, 0);        \
      im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(M_.x, X_[t].y, im[t], 0, 0, 0);        \
    }                                                                                     \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                      \
      re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(nay_, X_[t].y, re[t], 0, 0, 0);        \
      im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(M_.y, X_[t].x, im[t], 0, 0, 0);        \
    }                                                                                     \
  }
  cplxf mm[STG];
  cplxf xx[STG][NT];
#pragma unroll
  for (int s = 0; s < STG; ++s) SW_BSRF_LOAD(mm[s], xx[s], (s < KS ? s : 0));
  for (int ks = 0; ks < KS; ks += STG) {
#pragma unroll
    for (int s = 0; s < STG; ++s) {
      if (ks + s < KS) {                                            // KS need not divide by STG
        const int kn = (ks + s + STG < KS) ? ks + s + STG : ks + s;   // tail: harmless re-load
        __builtin_amdgcn_sched_barrier(0);
        SW_BSRF_MFMA(mm[s], xx[s]);
        __builtin_amdgcn_sched_barrier(0);
        SW_BSRF_LOAD(mm[s], xx[s], kn);
      }
    }
  }
#undef SW_BSRF_LOAD
#undef SW_BSRF_MFMA
  const int ot = tmap ? __builtin_amdgcn_readfirstlane(tmap[rt]) : rt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t row = (size_t)ot * 16 + 4 * (lane >> 4) + r;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const size_t off = row * nbp + c0 + t * 16 + (lane & 15);
      cplxf y = cmake(re[t][r], im[t][r]);
      if (MODE == 1) y = csub(B[off], y);
      if (MODE == 3) {
        cplxf o = X[off];
        cfma(o, w, csub(B[off], y));
        y = o;
      }
      Y[off] = y;
    }
  }
}

// Split-K twin for operators with few (tile, chunk) pairs and long rows (the dense coarsest inverse:
// 64 row tiles x 256 k-steps): the four waves of a workgroup share one (row tile, probe chunk), wave q
// takes the k-steps q, q+4, ..., the partial accumulators meet in LDS and wave 0 runs the epilogue.
template <int MODE, int NT>
__global__ __launch_bounds__(SW_BLOCK) void k_bsr_mfma_f32_sk(const cplxf* __restrict__ Ap,
                                                              const int* __restrict__ kcol, int KS,
                                                              int RT, const cplxf* __restrict__ X,
                                                              const cplxf* __restrict__ B,
                                                              cplxf* __restrict__ Y, int nbp, cplxf w,
                                                              const int* __restrict__ tmap) {
  __shared__ float red[3][NT * 8][64];
  const int lane = threadIdx.x & 63;
  const int q = threadIdx.x >> 6;
  const int rt = blockIdx.x % RT;
  const int cy = blockIdx.x / RT;
  const int c0 = cy * (16 * NT);
  const cplxf* a = Ap + (size_t)rt * KS * 64 + lane;
  const int* kc = kcol + (size_t)rt * KS;
  const cplxf* b = X + (size_t)(lane >> 4) * nbp + c0 + (lane & 15);
  sw_float4 re[NT], im[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    re[t] = sw_float4{0.f, 0.f, 0.f, 0.f};
    im[t] = re[t];
  }
  constexpr int STG = 4;
  cplxf mm[STG];
  cplxf xx[STG][NT];
#define SW_SKF_LOAD(S_, KSI)                                                  \
  {                                                                           \
    const int ks_ = ((KSI) < KS) ? (KSI) : q;                                 \
    mm[S_] = a[(size_t)ks_ * 64];                                             \
    const cplxf* bk_ = b + (size_t)kc[ks_] * nbp;                             \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) xx[S_][t] = bk_[t * 16];   \
  }
#pragma unroll
  for (int s = 0; s < STG; ++s) SW_SKF_LOAD(s, q + 4 * s);
  for (int ks = q; ks < KS; ks += 4 * STG) {
#pragma unroll
    for (int s = 0; s < STG; ++s) {
      if (ks + 4 * s < KS) {
        __builtin_amdgcn_sched_barrier(0);
        const float nay = -mm[s].y;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(mm[s].x, xx[s][t].x, re[t], 0, 0, 0);
          im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(mm[s].x, xx[s][t].y, im[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          re[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(nay, xx[s][t].y, re[t], 0, 0, 0);
          im[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(mm[s].y, xx[s][t].x, im[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        SW_SKF_LOAD(s, ks + 4 * s + 4 * STG);
      }
    }
  }
#undef SW_SKF_LOAD
  if (q > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[q - 1][t * 8 + r][lane] = re[t][r];
        red[q - 1][t * 8 + 4 + r][lane] = im[t][r];
      }
  }
  __syncthreads();
  if (q != 0) return;
  const int ot = tmap ? __builtin_amdgcn_readfirstlane(tmap[rt]) : rt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const size_t row = (size_t)ot * 16 + 4 * (lane >> 4) + r;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const size_t off = row * nbp + c0 + t * 16 + (lane & 15);
      cplxf y = cmake(re[t][r] + red[0][t * 8 + r][lane] + red[1][t * 8 + r][lane] + red[2][t * 8 + r][lane],
                      im[t][r] + red[0][t * 8 + 4 + r][lane] + red[1][t * 8 + 4 + r][lane] +
                          red[2][t * 8 + 4 + r][lane]);
      if (MODE == 1) y = csub(B[off], y);
      if (MODE == 3) {
        cplxf o = X[off];
        cfma(o, w, csub(B[off], y));
        y = o;
      }
      Y[off] = y;
    }
  }
}

// precision boundary of the single-precision preconditioner: dst = (CO) src, element-wise
template <class CI, class CO>
__global__ __launch_bounds__(SW_BLOCK) void k_cast(const CI* __restrict__ src, CO* __restrict__ dst,
                                                   size_t count) {
  const size_t i = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  if (i >= count) return;
  const CI v = src[i];
  CO o;
  o.x = (typename real_of<CO>::type)v.x;
  o.y = (typename real_of<CO>::type)v.y;
  dst[i] = o;
}

// ------------------------------------------------------------------------------------------
// Device-side construction of a block level's even-odd operators (sw_setup_eo_operators):
//   G = D_oo^-1,  F = A_eo G,  Hb = G A_oe,  S = D_ee - F A_oe
// as batched 16 x 16 complex algebra on site blocks in the kernels' own packed form: a block is four
// k-step tiles of 64 = 256 values, element (i, c) at [(c >> 2) * 64 + i + 16 * (c & 3)].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int blk_pos_row(int q) { return q & 15; }                       // packed position -> row
__device__ __forceinline__ int blk_pos_col(int q) { return 4 * (q >> 6) + ((q & 63) >> 4); }   // -> column

// inv[b] = src[b]^-1 (Gauss-Jordan with partial pivoting), one workgroup of 256 threads per block;
// soff / doff: element offsets of the source and destination blocks; info[0] |= 1 on a zero pivot
__global__ __launch_bounds__(256) void k_block_inverse(const cplx* __restrict__ src,
                                                       const long long* __restrict__ soff,
                                                       cplx* __restrict__ dst,
                                                       const long long* __restrict__ doff,
                                                       int* __restrict__ info) {
  __shared__ cplx a[16][33];       // [A | I], padded
  __shared__ cplx fac[16];
  __shared__ int piv;
  const int q = threadIdx.x;
  const int i = blk_pos_row(q), c = blk_pos_col(q);
  const cplx* s = src + soff[blockIdx.x];
  a[i][c] = s[q];
  a[i][16 + c] = cmake(i == c ? 1.0 : 0.0, 0.0);
  __syncthreads();
  for (int k = 0; k < 16; ++k) {
    if (q == 0) {
      int p = k;
      double best = a[k][k].x * a[k][k].x + a[k][k].y * a[k][k].y;
      for (int r = k + 1; r < 16; ++r) {
        const double m = a[r][k].x * a[r][k].x + a[r][k].y * a[r][k].y;
        if (m > best) {
          best = m;
          p = r;
        }
      }
      piv = p;
      if (best == 0.0) atomicOr(info, 1);
    }
    __syncthreads();
    const int p = piv;
    if (p != k && q < 32) {
      const cplx t = a[k][q];
      a[k][q] = a[p][q];
      a[p][q] = t;
    }
    __syncthreads();
    if (q < 16) fac[q] = a[q][k];             // column k before it is eliminated
    __syncthreads();
    const cplx pv = fac[k];
    const double d = pv.x * pv.x + pv.y * pv.y;
    const cplx ip = d > 0.0 ? cmake(pv.x / d, -pv.y / d) : cmake(0.0, 0.0);
    if (q < 32) a[k][q] = cmul(a[k][q], ip);
    __syncthreads();
    // rows r != k: a[r][:] -= fac[r] * a[k][:]   (512 entries, two per thread)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = q + 256 * e;
      const int r = idx >> 5, cc = idx & 31;
      if (r != k) {
        const cplx f = fac[r];
        const cplx t = cmul(f, a[k][cc]);
        a[r][cc] = csub(a[r][cc], t);
      }
    }
    __syncthreads();
  }
  dst[doff[blockIdx.x] + q] = a[i][16 + c];
}

// out[b] = (ioff[b] >= 0 ? init[ioff[b]] : 0) + sign * sum_{t in [ptr[b], ptr[b+1])} A[aoff[t]] B[boff[t]]
// (16 x 16 complex products of packed blocks), one workgroup of 256 threads per output block
__global__ __launch_bounds__(256) void k_block_products(const int* __restrict__ ptr,
                                                        const long long* __restrict__ aoff,
                                                        const long long* __restrict__ boff,
                                                        const cplx* __restrict__ A,
                                                        const cplx* __restrict__ B,
                                                        const long long* __restrict__ ioff,
                                                        const cplx* __restrict__ init, double sign,
                                                        cplx* __restrict__ out,
                                                        const long long* __restrict__ ooff) {
  __shared__ cplx as[16][17], bs[16][17];
  const int q = threadIdx.x;
  const int i = blk_pos_row(q), c = blk_pos_col(q);
  const int b = blockIdx.x;
  cplx acc = cmake(0.0, 0.0);
  for (int t = ptr[b]; t < ptr[b + 1]; ++t) {
    as[i][c] = A[aoff[t] + q];
    bs[i][c] = B[boff[t] + q];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m) cfma(acc, as[i][m], bs[m][c]);
    __syncthreads();
  }
  cplx o = cmake(sign * acc.x, sign * acc.y);
  if (ioff[b] >= 0) o = cadd(o, init[ioff[b] + q]);
  out[ooff[b] + q] = o;
}

// ------------------------------------------------------------------------------------------
// In-place dense inverse (gj_invert: coarsest operator, directly solved levels): Gauss-Jordan with
// partial pivoting on the row-major [n][n] matrix, four small launches per pivot step.  Unblocked, every
// step is a rank-1 update of the whole matrix (n^3 complex updates = 0.35 s of memory traffic at n = 4096);
// blocked, the steps of a panel of nb columns touch the panel only and the rest of the matrix follows in one
// rank-nb update on the matrix cores (k_gj_block_rows, k_gj_panel_to_bsr, k_bsr_mfma3).
//   step k:  p = argmax_{r >= k} |a[r][k]| ;  rows k <-> p ;  c_r = a[r][k], column k := e_k ;
//            row k *= 1 / pivot ;  a[r][:] -= c_r a[k][:]  (r != k) ;   afterwards the row swaps are
//   undone on the columns in reverse order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_gj_pivot(const cplx* __restrict__ A, int n, int k,
                                                   int* __restrict__ pivs, cplx* __restrict__ pvinv,
                                                   int* __restrict__ info) {
  __shared__ double bm[1024];
  __shared__ int bi[1024];
  const int t = threadIdx.x;
  double best = -1.0;
  int arg = k;
  for (int r = k + t; r < n; r += 1024) {
    const cplx v = A[(size_t)r * n + k];
    const double m = v.x * v.x + v.y * v.y;
    if (m > best) {
      best = m;
      arg = r;
    }
  }
  bm[t] = best;
  bi[t] = arg;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s && (bm[t + s] > bm[t] || (bm[t + s] == bm[t] && bi[t + s] < bi[t]))) {
      bm[t] = bm[t + s];
      bi[t] = bi[t + s];
    }
    __syncthreads();
  }
  if (t == 0) {
    const int p = bi[0];
    pivs[k] = p;
    const cplx v = A[(size_t)p * n + k];
    const double d = v.x * v.x + v.y * v.y;
    if (d > 0.0) {
      *pvinv = cmake(v.x / d, -v.y / d);
    } else {
      *pvinv = cmake(0.0, 0.0);
      atomicOr(info, 1);
    }
  }
}

// (the three row kernels work on the column window [c0, c0 + cn): the whole matrix in the unblocked
// algorithm, the current panel in the blocked one)
__global__ __launch_bounds__(SW_BLOCK) void k_gj_swap_rows(cplx* __restrict__ A, int n, int k,
                                                           const int* __restrict__ pivs, int c0, int cn) {
  const int t = blockIdx.x * SW_BLOCK + threadIdx.x;
  const int p = pivs[k];
  if (t >= cn || p == k) return;
  const int c = c0 + t;
  const cplx v = A[(size_t)k * n + c];
  A[(size_t)k * n + c] = A[(size_t)p * n + c];
  A[(size_t)p * n + c] = v;
}

// thread t: as a row, colk[t] = a[t][k] and a[t][k] := 0 (t != k); then, as a column of the window,
// a[k][c0 + t] *= 1/pivot, where a[k][k] counts as 1 (column k := e_k)
__global__ __launch_bounds__(SW_BLOCK) void k_gj_column_and_scale(cplx* __restrict__ A, int n, int k,
                                                                  const cplx* __restrict__ pvinv,
                                                                  cplx* __restrict__ colk, int c0, int cn) {
  const int t = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (t < n && t != k) {
    colk[t] = A[(size_t)t * n + k];
    A[(size_t)t * n + k] = cmake(0.0, 0.0);
  }
  if (t < cn) {
    const int c = c0 + t;
    A[(size_t)k * n + c] = (c == k) ? *pvinv : cmul(A[(size_t)k * n + c], *pvinv);
  }
}

// a[r][c] -= colk[r] * a[k][c] for r != k, c in the window; one wave per row segment of 64 columns
__global__ __launch_bounds__(SW_BLOCK) void k_gj_update(cplx* __restrict__ A, int n, int k,
                                                        const cplx* __restrict__ colk, int c0, int cn) {
  const int t = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r0 = (blockIdx.y * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16;
  if (t >= cn) return;
  const int c = c0 + t;
  const cplx rk = A[(size_t)k * n + c];
#pragma unroll 4
  for (int r = r0; r < r0 + 16 && r < n; ++r) {
    if (r == k) continue;
    cplx v = A[(size_t)r * n + c];
    const cplx f = colk[r];
    cfma(v, cmake(-f.x, -f.y), rk);
    A[(size_t)r * n + c] = v;
  }
}

// The pivot step of the blocked algorithm in two launches (panel of at most 64 columns).
// k_gj_pivot_panel, one workgroup: p = argmax_{r >= k} |a[r][k]|; rows k <-> p inside the panel; row k of
// the panel *= 1 / pivot, with a[k][k] := 1 / pivot.
__global__ __launch_bounds__(1024) void k_gj_pivot_panel(cplx* __restrict__ A, int n, int k, int c0, int cn,
                                                         int* __restrict__ pivs, int* __restrict__ info) {
  __shared__ double bm[1024];
  __shared__ int bi[1024];
  const int t = threadIdx.x;
  double best = -1.0;
  int arg = k;
  for (int r = k + t; r < n; r += 1024) {
    const cplx v = A[(size_t)r * n + k];
    const double m = v.x * v.x + v.y * v.y;
    if (m > best) {
      best = m;
      arg = r;
    }
  }
  bm[t] = best;
  bi[t] = arg;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s && (bm[t + s] > bm[t] || (bm[t + s] == bm[t] && bi[t + s] < bi[t]))) {
      bm[t] = bm[t + s];
      bi[t] = bi[t + s];
    }
    __syncthreads();
  }
  const int p = bi[0];
  const cplx pv = A[(size_t)p * n + k];
  const double d = pv.x * pv.x + pv.y * pv.y;
  const cplx pinv = d > 0.0 ? cmake(pv.x / d, -pv.y / d) : cmake(0.0, 0.0);
  __syncthreads();                       // every thread has read the pivot before the rows move
  if (t == 0) {
    pivs[k] = p;
    if (!(d > 0.0)) atomicOr(info, 1);
  }
  if (t < cn) {
    const int c = c0 + t;
    const cplx vp = A[(size_t)p * n + c];
    if (p != k) A[(size_t)p * n + c] = A[(size_t)k * n + c];
    A[(size_t)k * n + c] = (c == k) ? pinv : cmul(vp, pinv);
  }
}

// k_gj_update_panel: a[r][c] -= a[r][k] a[k][c] for r != k and the panel's columns c != k, a[r][k] := -a[r][k] a[k][k]
// (column k := e_k before the update).  One wave = 16 rows x the panel (cn <= 64: one lane per column, so the
// lanes of a wave read a[r][k] before the lane of column k overwrites it).
__global__ __launch_bounds__(SW_BLOCK) void k_gj_update_panel(cplx* __restrict__ A, int n, int k, int c0, int cn) {
  const int t = threadIdx.x & 63;
  const int r0 = (blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16;
  const int c = c0 + (t < cn ? t : 0);
  const cplx rk = A[(size_t)k * n + c];
#pragma unroll 4
  for (int r = r0; r < r0 + 16 && r < n; ++r) {
    if (r == k) continue;
    const cplx f = A[(size_t)r * n + k];
    cplx v = (c == k) ? cmake(0.0, 0.0) : A[(size_t)r * n + c];
    cfma(v, cmake(-f.x, -f.y), rk);
    if (t < cn) A[(size_t)r * n + c] = v;
  }
}

// Blocked Gauss-Jordan (gj_invert): after the nb columns k0 .. k0 + nb - 1 have been eliminated inside their
// panel, the panel holds N = G E (the composite transformation applied to the unit columns), and the other
// columns c follow as  a[:, c] <- a[:, c] with the pivot rows zeroed  +  N a[k0 .. k0 + nb, c].
// k_gj_block_rows, one thread per column: the panel's nb row swaps, then T[b][c] = a[k0 + b][c] and
// a[k0 + b][c] := 0 (columns of the panel: T = 0, nothing else).
__global__ __launch_bounds__(SW_BLOCK) void k_gj_block_rows(cplx* __restrict__ A, int n, int k0, int nb,
                                                            const int* __restrict__ pivs, cplx* __restrict__ T) {
  const int c = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (c >= n) return;
  if (c >= k0 && c < k0 + nb) {
    for (int b = 0; b < nb; ++b) T[(size_t)b * n + c] = cmake(0.0, 0.0);
    return;
  }
  for (int j = k0; j < k0 + nb; ++j) {
    const int p = pivs[j];
    if (p != j) {
      const cplx v = A[(size_t)j * n + c];
      A[(size_t)j * n + c] = A[(size_t)p * n + c];
      A[(size_t)p * n + c] = v;
    }
  }
  for (int b = 0; b < nb; ++b) {
    T[(size_t)b * n + c] = A[(size_t)(k0 + b) * n + c];
    A[(size_t)(k0 + b) * n + c] = cmake(0.0, 0.0);
  }
}

// -N, the negated panel a[:, k0 .. k0 + nb), in MFMA block-row form (KS = nb / 4 four-column groups per
// 16-row tile, k_dense_to_bsr's tile layout), so that k_bsr_mfma3's residual mode does a <- a - (-N) T
__global__ __launch_bounds__(SW_BLOCK) void k_gj_panel_to_bsr(const cplx* __restrict__ A, int n, int k0, int nb,
                                                              cplx* __restrict__ vals, int* __restrict__ kcol) {
  const int lane = threadIdx.x & 63;
  const int KS = nb >> 2;
  const size_t item = (size_t)blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= (size_t)(n >> 4) * KS) return;
  const int rt = (int)(item / KS), ks = (int)(item - (size_t)rt * KS);
  const cplx v = A[((size_t)rt * 16 + (lane & 15)) * n + k0 + 4 * ks + (lane >> 4)];
  vals[item * 64 + lane] = cmake(-v.x, -v.y);
  if (lane == 0) kcol[item] = 4 * ks;
}

// undo the row swaps on the columns, last swap first; one thread per row
__global__ __launch_bounds__(SW_BLOCK) void k_gj_unpermute(cplx* __restrict__ A, int n,
                                                           const int* __restrict__ pivs) {
  const int r = blockIdx.x * SW_BLOCK + threadIdx.x;
  if (r >= n) return;
  cplx* row = A + (size_t)r * n;
  for (int k = n - 1; k >= 0; --k) {
    const int p = pivs[k];
    if (p != k) {
      const cplx t = row[k];
      row[k] = row[p];
      row[p] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Pack / unpack between the reference's host layout ([probe][natural index], probe-major) and
// the engine layout ([internal row][probe]).  rowmap[natural] = internal row (NULL: identity).
// ------------------------------------------------------------------------------------------
// probes: int8 code -> complex.  +-1 are the reference's Rademacher (Z2) entries; the build-only Z4
// option adds +-2 meaning +-i (BASELINE config 1; not in the reference, utils.py:213-216)
__global__ __launch_bounds__(SW_BLOCK) void k_pack_i8(const int8_t* __restrict__ src, int nb,
                                                      int n, const int* __restrict__ rowmap,
                                                      cplx* __restrict__ dst, int nbp) {
  // tile of 64 natural indices x 64 probes through LDS so both sides are coalesced
  __shared__ int8_t tile[64][65];
  const int i0 = blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int jj = ty; jj < 64; jj += 4) {
    const int j = j0 + jj, i = i0 + tx;
    tile[jj][tx] = (j < nb && i < n) ? src[(size_t)j * n + i] : (int8_t)0;
  }
  __syncthreads();
  for (int ii = ty; ii < 64; ii += 4) {
    const int i = i0 + ii;
    if (i < n) {
      const size_t row = rowmap ? (size_t)rowmap[i] : (size_t)i;
      const int v = tile[tx][ii];
      dst[row * nbp + j0 + tx] = (v == 2 || v == -2) ? cmake(0.0, (double)(v / 2)) : cmake((double)v, 0.0);
    }
  }
}

__global__ __launch_bounds__(SW_BLOCK) void k_pack_c(const cplx* __restrict__ src, int nb, int n,
                                                     const int* __restrict__ rowmap,
                                                     cplx* __restrict__ dst, int nbp) {
  __shared__ cplx tile[64][65];
  const int i0 = blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int jj = ty; jj < 64; jj += 4) {
    const int j = j0 + jj, i = i0 + tx;
    tile[jj][tx] = (j < nb && i < n) ? src[(size_t)j * n + i] : cmake(0.0, 0.0);
  }
  __syncthreads();
  for (int ii = ty; ii < 64; ii += 4) {
    const int i = i0 + ii;
    if (i < n) {
      const size_t row = rowmap ? (size_t)rowmap[i] : (size_t)i;
      dst[row * nbp + j0 + tx] = tile[tx][ii];
    }
  }
}

__global__ __launch_bounds__(SW_BLOCK) void k_unpack_c(const cplx* __restrict__ src, int nbp,
                                                       int n, const int* __restrict__ rowmap,
                                                       cplx* __restrict__ dst, int nb) {
  __shared__ cplx tile[64][65];
  const int i0 = blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int ii = ty; ii < 64; ii += 4) {
    const int i = i0 + ii;
    if (i < n) {
      const size_t row = rowmap ? (size_t)rowmap[i] : (size_t)i;
      tile[ii][tx] = src[row * nbp + j0 + tx];
    }
  }
  __syncthreads();
  for (int jj = ty; jj < 64; jj += 4) {
    const int j = j0 + jj, i = i0 + tx;
    if (j < nb && i < n) dst[(size_t)j * n + i] = tile[tx][jj];
  }
}

// ------------------------------------------------------------------------------------------
// Device-side probe generation: the reference's MT19937 stream (np.random.randint(2, size=n),
// utils.py:213-216,255-258), bit-exact, started anywhere in the stream.
//   window = the 624 raw (untempered) words w[p .. p+623] at stream position p (draw index).
//   jump   : window at p + J:  W'[k] = XOR_{i : g_i = 1} w[p + i + k],  g = x^J mod phi (host,
//            sw_mt19937.cpp); the 20560-word table w lives in LDS (82 KB), thread k accumulates
//            its output word with wave-uniform polynomial bits from the scalar path.
//   generate: the word recurrence w[t+624] = w[t+397] ^ twist(w[t], w[t+1]) is parallel over 227
//            consecutive t, i.e. one 624-word block = three barrier-separated phases.
// One workgroup = one segment of the batch: it jumps from the batch's base window to its own
// start (polynomial s of the family x^(s*segdraws)), then walks its segment block by block,
// tempering each word and storing the probe entries as int8 codes in the layout an uploaded
// probe batch has ([probe][natural index]; +-1 = Z2, +-1/+-2 = Z4), four entries per store.
// ------------------------------------------------------------------------------------------
#define SW_MT_N 624
#define SW_MT_M 397
#define SW_MT_DEG 19937
#define SW_MT_TABLE (SW_MT_DEG + SW_MT_N)   // w[i + k], i < DEG, k < 624

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

#define SW_MT_BLOCK 1024   // 16 waves: the convolution is LDS-latency bound, its i-range is split 4 ways

// w[0..623] valid on entry; fills w[624 .. SW_MT_TABLE-1].  All threads of the block must call.
__device__ __forceinline__ void mt_extend_table(uint32_t* w) {
  const int tid = threadIdx.x;
  for (int base = 0; base + SW_MT_N < SW_MT_TABLE; base += 227) {
    const int t = base + tid;
    if (tid < 227 && t + SW_MT_N < SW_MT_TABLE) w[t + SW_MT_N] = w[t + SW_MT_M] ^ mt_twist(w[t], w[t + 1]);
    __syncthreads();
  }
}

// Jumped window into dst[0..623] (LDS or global).  Thread (q, t) = (tid >> 8, tid & 255) takes the
// polynomial words [156 q, 156 q + 156) and the output words k = t, t + 256, t + 512; the four
// partial sums meet in `red` ([4][3][256] words of LDS).  All 1024 threads must call.
__device__ __forceinline__ void mt_convolve(const uint32_t* w, const uint32_t* __restrict__ poly,
                                            uint32_t* red, uint32_t* dst) {
  const int tid = threadIdx.x;
  const int q = tid >> 8, t = tid & 255;
  uint32_t a0 = 0u, a1 = 0u, a2 = 0u;
  const bool third = (t + 512) < SW_MT_N;
  const int w0 = __builtin_amdgcn_readfirstlane(q * (SW_MT_N / 4));
  for (int wi = w0; wi < w0 + SW_MT_N / 4; ++wi) {
    uint32_t bits = __builtin_amdgcn_readfirstlane(poly[wi]);
    const uint32_t* wb = w + wi * 32 + t;
    while (bits) {
      const int b = __builtin_ctz(bits);
      bits &= bits - 1;
      a0 ^= wb[b];
      a1 ^= wb[b + 256];
      if (third) a2 ^= wb[b + 512];
    }
  }
  red[(q * 3 + 0) * 256 + t] = a0;
  red[(q * 3 + 1) * 256 + t] = a1;
  red[(q * 3 + 2) * 256 + t] = a2;
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int k = t + 256 * r;
      if (k < SW_MT_N)
        dst[k] = red[(0 * 3 + r) * 256 + t] ^ red[(1 * 3 + r) * 256 + t] ^
                 red[(2 * 3 + r) * 256 + t] ^ red[(3 * 3 + r) * 256 + t];
    }
  }
  __syncthreads();
}

// window_out = window_in advanced by the jump polynomial `poly` (one workgroup)
__global__ __launch_bounds__(SW_MT_BLOCK) void k_mt_jump(const uint32_t* __restrict__ win_in,
                                                         const uint32_t* __restrict__ poly,
                                                         uint32_t* __restrict__ win_out) {
  __shared__ uint32_t w[SW_MT_TABLE + 8];
  __shared__ uint32_t red[4 * 3 * 256];
  const int tid = threadIdx.x;
  if (tid < SW_MT_N) w[tid] = win_in[tid];
  __syncthreads();
  mt_extend_table(w);
  mt_convolve(w, poly, red, win_out);
}

// kind 1: Z2 (entry = 2 (y & 1) - 1);  kind 2: Z4 (y & 3 -> 1, i, -1, -i  ==  codes 1, 2, -1, -2)
__device__ __forceinline__ int mt_code(uint32_t y, int kind) {
  if (kind == 1) return 2 * (int)(y & 1u) - 1;
  const int q = (int)(y & 3u);
  return (q == 0) ? 1 : (q == 1) ? 2 : (q == 2) ? -1 : -2;
}

// grid = number of segments; segment s covers draws [s*segdraws, min(total, (s+1)*segdraws)) of the
// batch that starts at `win` ; polys[(s-1)*624 ..] = x^(s*segdraws) mod phi.  segdraws % 4 == 0.
__global__ __launch_bounds__(SW_MT_BLOCK) void k_mt_generate(const uint32_t* __restrict__ win,
                                                             const uint32_t* __restrict__ polys,
                                                             unsigned long long segdraws,
                                                             unsigned long long total, int kind,
                                                             int8_t* __restrict__ out) {
  __shared__ uint32_t w[SW_MT_TABLE + 8];
  __shared__ uint32_t red[4 * 3 * 256];
  __shared__ __attribute__((aligned(16))) uint32_t st[SW_MT_N];
  const int tid = threadIdx.x;
  const int s = blockIdx.x;
  if (tid < SW_MT_N) w[tid] = win[tid];
  __syncthreads();
  if (s > 0) {
    mt_extend_table(w);
    mt_convolve(w, polys + (size_t)(s - 1) * SW_MT_N, red, st);
  } else {
    if (tid < SW_MT_N) st[tid] = w[tid];
    __syncthreads();
  }
  // the walk along the segment needs 227 threads: waves 4..15 end here (whole waves; s_barrier
  // waits only on the waves of the workgroup that are still running, CDNA ISA "S_BARRIER")
  if (tid >= 256) return;
  const unsigned long long d0 = (unsigned long long)s * segdraws;
  const unsigned long long d1 = (d0 + segdraws < total) ? d0 + segdraws : total;
  for (unsigned long long d = d0; d < d1; d += SW_MT_N) {
    // emit the block: thread q < 156 takes words 4q .. 4q+3
    if (tid < SW_MT_N / 4) {
      const unsigned long long e = d + 4ull * tid;
      if (e < d1) {
        const uint4 v = *reinterpret_cast<const uint4*>(&st[4 * tid]);
        const int c0 = mt_code(mt_temper(v.x), kind), c1 = mt_code(mt_temper(v.y), kind);
        const int c2 = mt_code(mt_temper(v.z), kind), c3 = mt_code(mt_temper(v.w), kind);
        if (e + 4 <= d1) {
          const uint32_t pk = (uint32_t)(c0 & 0xff) | ((uint32_t)(c1 & 0xff) << 8) |
                              ((uint32_t)(c2 & 0xff) << 16) | ((uint32_t)(c3 & 0xff) << 24);
          *reinterpret_cast<uint32_t*>(out + e) = pk;
        } else {
          out[e] = (int8_t)c0;
          if (e + 1 < d1) out[e + 1] = (int8_t)c1;
          if (e + 2 < d1) out[e + 2] = (int8_t)c2;
        }
      }
    }
    if (d + SW_MT_N >= d1) break;
    // next block in place: three phases of <= 227 independent words
    uint32_t a = 0, b = 0, c = 0;
    if (tid < 227) { a = st[tid]; b = st[tid + 1]; c = st[tid + SW_MT_M]; }
    __syncthreads();
    if (tid < 227) st[tid] = c ^ mt_twist(a, b);
    __syncthreads();
    if (tid < 227) { a = st[227 + tid]; b = st[228 + tid]; c = st[tid]; }
    __syncthreads();
    if (tid < 227) st[227 + tid] = c ^ mt_twist(a, b);
    __syncthreads();
    if (tid < 170) { a = st[454 + tid]; b = st[(tid == 169) ? 0 : 455 + tid]; c = st[227 + tid]; }
    __syncthreads();
    if (tid < 170) st[454 + tid] = c ^ mt_twist(a, b);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Per-probe scalars of the batched flexible GMRES (one thread per probe).
// Storage: H [(m+1)][m][nbp], cs [m][nbp] (.x), sn [m][nbp], g [(m+1)][nbp].
// The three updates below (start of a cycle, Hessenberg column, true-residual check) each consume the
// result of ONE reduction; they run either as the small kernel k_fg_tail or -- option fused_reduce -- in
// the tail of the reducing launch itself (FgTail, reduce_and_tail): the workgroup that completes a
// probe chunk's reduction applies the update to that chunk's 64 columns.
// Two tolerances: `tol` is the reference's (a probe's reported iteration count is the first iteration
// at which its residual passes it, multigrid.py:347-366), `tol_stop` <= tol is what the batch is
// iterated to (engine option stop_factor; tol_stop = tol reproduces the reference's stopping point).
// ------------------------------------------------------------------------------------------
struct FgScalars {
  cplx* H;
  cplx* cs;
  cplx* sn;
  cplx* g;
  cplx* y;        // [m][nbp]
  cplx* normb;    // [nbp] .x
  cplx* relres;   // [nbp] .x
  cplx* svec;     // [(m+1)][nbp] .x: scale of basis vector k, v_k = svec_k * vtilde_k (0: frozen)
  cplx* ys;       // [m][nbp] y_k * svec_k: coefficients of the update x += sum_k ys_k ztilde_k
  int* iters;     // [nbp] iteration at which the probe first met tol (-1: not yet)
  int m;
  int nbp;
};

#define SW_TAIL_NONE 0
#define SW_TAIL_BEGIN 1
#define SW_TAIL_HESS 2
#define SW_TAIL_VERIFY 3
#define SW_TAIL_GRAM 4
#define SW_GRAM_MAXL 4       // longest restart cycle of the Gram-matrix form (5 vectors, 15 inner products)
struct FgTail {
  int kind;
  FgScalars s;
  int j;               // HESS: column of the Hessenberg matrix
  const cplx* h1;      // HESS: raw dots of pass 1;  BEGIN / VERIFY: d[col].x = the squared norm
  const cplx* h2;      // HESS: raw dots of the second Gram-Schmidt pass, or NULL
  const cplx* nrm2;    // HESS: |vtilde_{j+1}|^2 (pyth: |A ztilde_j|^2)
  double tol, tol_stop;
  int iter_base, pyth, first_cycle;
  int* notconv;        // HESS / VERIFY: counter of the probes still above tol_stop (zero on entry)
};

// start of a cycle: beta = sqrt(d[col].x);  first cycle also fixes normb
__device__ __forceinline__ void fg_begin_col(const FgScalars& s, const cplx* __restrict__ d,
                                             int first_cycle, int col) {
  const double beta = sqrt(fmax(d[col].x, 0.0));
  if (first_cycle) {
    s.normb[col] = cmake(beta, 0.0);
    s.iters[col] = (beta > 0.0) ? -1 : 0;
    s.relres[col] = cmake(beta > 0.0 ? 1.0 : 0.0, 0.0);
  }
  s.g[col] = cmake(beta, 0.0);
  s.svec[col] = cmake(beta > 0.0 ? 1.0 / beta : 0.0, 0.0);   // vtilde_0 = r itself
}

// Column j of the Hessenberg matrix.  The Krylov basis is kept UNNORMALISED in memory: the stored
// vtilde_k and the true orthonormal v_k differ by a per-probe scale, v_k = svec_k vtilde_k (vtilde_0
// is the residual itself, vtilde_{k+1} the orthogonalised A M vtilde_k as multiaxpy leaves it), so no
// pass over the vectors is spent on normalisation; the multigrid cycle and A are linear, so
// ztilde_k = M vtilde_k and z_k = svec_k ztilde_k.  With the raw dots d_k = vtilde_k^H (A ztilde_j)
// (d1 + d2: two Gram-Schmidt passes) and nrm2 = |vtilde_{j+1}|^2:
//   h_{k,j} = svec_k svec_j d_k,   h_{j+1,j} = svec_j sqrt(nrm2),   svec_{j+1} = 1/sqrt(nrm2)
// (the orthogonalisation coefficients svec_k^2 d_k come from the reduction).
__device__ __forceinline__ void fg_hess_col(const FgScalars& s, int j, const cplx* __restrict__ h1,
                                            const cplx* __restrict__ h2, const cplx* __restrict__ nrm2,
                                            double tol, double tol_stop, int iter_base, int pyth,
                                            int* notconv, int col) {
  const int m = s.m, nbp = s.nbp;
  const double sj = s.svec[(size_t)j * nbp + col].x;
  // apply the previous rotations
  const double s0 = s.svec[col].x * sj;
  cplx hk = h2 ? cadd(h1[col], h2[col]) : h1[col];     // h2 == NULL: one Gram-Schmidt pass
  hk = cmake(s0 * hk.x, s0 * hk.y);
  double hcol2 = hk.x * hk.x + hk.y * hk.y;      // |h_{0..j,j}|^2 before the rotations
  for (int k = 0; k < j; ++k) {
    const double sk1 = s.svec[(size_t)(k + 1) * nbp + col].x * sj;
    cplx hk1 = h1[(size_t)(k + 1) * nbp + col];
    if (h2) hk1 = cadd(hk1, h2[(size_t)(k + 1) * nbp + col]);
    hk1 = cmake(sk1 * hk1.x, sk1 * hk1.y);
    hcol2 = fma(hk1.x, hk1.x, fma(hk1.y, hk1.y, hcol2));
    const double c = s.cs[(size_t)k * nbp + col].x;
    const cplx sn = s.sn[(size_t)k * nbp + col];
    // [ c  sn ; -conj(sn)  c ]
    cplx t = cmake(c * hk.x, c * hk.y);
    cfma(t, sn, hk1);
    cplx u = cmake(c * hk1.x, c * hk1.y);
    cfma(u, cmake(-sn.x, sn.y), hk);
    s.H[((size_t)k * m + j) * nbp + col] = t;
    hk = u;
  }
  // h_{j+1,j} = hn.  pyth == 0: nrm2 = |vtilde_{j+1}|^2 of the orthogonalised vector.  pyth == 1
  // (last step of a restart cycle, whose vtilde_{j+1} is never used and therefore never formed):
  // nrm2 = |A ztilde_j|^2 BEFORE the orthogonalisation, and with one classical Gram-Schmidt pass
  // |h_{j+1,j}|^2 = svec_j^2 |A ztilde_j|^2 - sum_k |h_{k,j}|^2; a remainder below 1e-12 of the
  // total is round-off of that subtraction and counts as breakdown (the solve is then settled by
  // the true-residual verification)
  double wn, hn;
  if (pyth) {
    const double tot = sj * sj * fmax(nrm2[col].x, 0.0);
    double h2v = tot - hcol2;
    if (!(h2v > 1.0e-12 * tot)) h2v = 0.0;
    hn = sqrt(h2v);
    wn = (sj > 0.0) ? hn / sj : 0.0;
  } else {
    wn = sqrt(fmax(nrm2[col].x, 0.0));
    hn = sj * wn;
  }
  // new rotation annihilating h_{j+1,j} = hn
  const double habs = sqrt(hk.x * hk.x + hk.y * hk.y);
  const double dnm = sqrt(habs * habs + hn * hn);
  double c;
  cplx sn;
  if (dnm == 0.0) {
    c = 1.0;
    sn = cmake(0.0, 0.0);
  } else if (habs == 0.0) {
    c = 0.0;
    sn = cmake(1.0, 0.0);
  } else {
    c = habs / dnm;
    const double f = hn / (habs * dnm);
    sn = cmake(hk.x * f, hk.y * f);  // (a/|a|) * conj(b)/d, b = hn real
  }
  s.cs[(size_t)j * nbp + col] = cmake(c, 0.0);
  s.sn[(size_t)j * nbp + col] = sn;
  cplx hjj = cmake(c * hk.x, c * hk.y);
  cfma(hjj, sn, cmake(hn, 0.0));
  s.H[((size_t)j * m + j) * nbp + col] = hjj;
  const cplx gj = s.g[(size_t)j * nbp + col];
  const cplx gn = cmul(cmake(-sn.x, sn.y), gj);
  s.g[(size_t)(j + 1) * nbp + col] = gn;
  s.g[(size_t)j * nbp + col] = cmake(c * gj.x, c * gj.y);
  const double nb_ = s.normb[col].x;
  const double rr = (nb_ > 0.0) ? sqrt(gn.x * gn.x + gn.y * gn.y) / nb_ : 0.0;
  const bool dead = (hn == 0.0 && hcol2 == 0.0);     // frozen earlier: w = 0, rr is meaningless
  if (!dead) s.relres[col] = cmake(rr, 0.0);
  if (s.iters[col] < 0 && rr < tol) s.iters[col] = iter_base + j + 1;
  if (!dead && !(rr < tol_stop)) atomicAdd(notconv, 1);
  // Next basis vector v_{j+1} = w / h_{j+1,j}.  Probes that have met tol_stop keep iterating in
  // lockstep with the batch (free extra accuracy) until they are two orders below it; then, or at
  // (happy) breakdown h_{j+1,j} <= 1e-14 |A z_j| where the remainder is round-off that 1/h would
  // blow up to a unit vector, the probe is FROZEN: scale 0 makes v_{j+1}, hence z, w and every
  // later Hessenberg column of this probe exactly zero, and k_fg_solve gives y = 0 for them.
  const bool frozen = (s.iters[col] >= 0 && rr < 1.0e-2 * tol_stop) ||
                      (hn * hn <= 1.0e-28 * (hcol2 + hn * hn));
  s.svec[(size_t)(j + 1) * nbp + col] = cmake((hn > 0.0 && !frozen) ? 1.0 / wn : 0.0, 0.0);
}

// true-residual check after a solve: d[col].x = ||b - A x||^2; probes above tol lose their recorded
// iteration (it is set again when they converge in a later cycle), probes above tol_stop are counted
__device__ __forceinline__ void fg_verify_col(const FgScalars& s, const cplx* __restrict__ d, double tol,
                                              double tol_stop, int* notconv, int col) {
  const double nb_ = s.normb[col].x;
  const double rr = (nb_ > 0.0) ? sqrt(fmax(d[col].x, 0.0)) / nb_ : 0.0;
  s.relres[col] = cmake(rr, 0.0);
  if (!(rr < tol)) s.iters[col] = -1;
  if (!(rr < tol_stop)) atomicAdd(notconv, 1);
}

// Restart cycle in Gram-matrix form (fgmres_eo_gram).  The cycle's L directions z_0 .. z_{L-1} and their
// images w_j = S z_j are built WITHOUT any orthogonalisation (z_0 = M r_0, z_{j+1} = M w_j: the same Krylov
// space as the Arnoldi process spans); ONE pass over [r_0, w_0 .. w_{L-1}] then yields all their inner
// products G[a][b] = u_a^H u_b (a <= b, u_0 = r_0, u_{a+1} = w_a), and the minimal-residual combination
// x += sum_j y_j z_j follows per probe from the normal equations (W^H W) y = W^H r_0 by a Cholesky
// factorisation C C^H of the L x L matrix: forward substitution t = C^-1 W^H r_0 gives the residual norm of
// EVERY nested sub-cycle for free (|r_j|^2 = |r_0|^2 - sum_{k<=j} |t_k|^2: the probe's iteration count is the
// first j that passes tol, as with Givens rotations), back substitution gives y.  The basis is nearly
// collinear (S M ~ I - E, |E| ~ 0.1: cond(W^H W) ~ 1e6 for L = 3), which costs the normal equations ~1e-10 of
// relative accuracy in y -- irrelevant for a cycle that reduces the residual by ~1e-3 and is followed by a
// TRUE residual; a pivot below 1e-13 of its diagonal entry truncates the cycle there.
//   d[k][col]: the reduced inner products, k = a NV - a (a - 1) / 2 + (b - a), NV = L + 1.
// Outputs: ys[j][col] = y_j (update coefficients), relres = |r_L| / |b|, g[0] = |r_0| / |b|.
// init != 0 (a cycle from a zero guess that is a whole solve of its own, e.g. the K-cycle's inner iteration):
// r_0 is the right-hand side, its norm becomes normb.
__device__ __forceinline__ void fg_gram_col(const FgScalars& s, int L, const cplx* __restrict__ d, double tol,
                                            double tol_stop, int iter_base, int* notconv, int init, int col) {
  const int nbp = s.nbp, NV = L + 1;
  auto G = [&](int a, int b) -> cplx { return d[(size_t)(a * NV - (a * (a - 1)) / 2 + (b - a)) * nbp + col]; };
  const double g00 = fmax(G(0, 0).x, 0.0);
  if (init) {
    s.normb[col] = cmake(sqrt(g00), 0.0);
    s.iters[col] = (g00 > 0.0) ? -1 : 0;
  }
  const double nb_ = s.normb[col].x;
  const double rr0 = (nb_ > 0.0) ? sqrt(g00) / nb_ : 0.0;
  s.g[col] = cmake(rr0, 0.0);
  cplx C[SW_GRAM_MAXL][SW_GRAM_MAXL];
  cplx t[SW_GRAM_MAXL];
  double res2 = g00;
  int Leff = 0;
  // a probe whose residual is already two orders below the stopping tolerance (or has no right-hand side)
  // takes no further part: y = 0
  const bool done = !(nb_ > 0.0) || (s.iters[col] >= 0 && rr0 < 1.0e-2 * tol_stop) || g00 == 0.0;
  double rr = rr0;
  if (!done) {
#pragma unroll
    for (int j = 0; j < SW_GRAM_MAXL; ++j) {
      if (j >= L) break;
      // row j of the Cholesky factor of M[a][b] = w_a^H w_b = G(a + 1, b + 1)
      double dj = G(j + 1, j + 1).x;
      const double mjj = dj;
#pragma unroll
      for (int k = 0; k < SW_GRAM_MAXL; ++k) {
        if (k >= j) break;
        // M[j][k] = conj(M[k][j]) = conj(G(k + 1, j + 1))
        cplx v = G(k + 1, j + 1);
        v = cmake(v.x, -v.y);
#pragma unroll
        for (int i = 0; i < SW_GRAM_MAXL; ++i) {
          if (i >= k) break;
          // v -= C[j][i] conj(C[k][i])
          const cplx a_ = C[j][i], b_ = C[k][i];
          v = cmake(v.x - (a_.x * b_.x + a_.y * b_.y), v.y - (a_.y * b_.x - a_.x * b_.y));
        }
        const double ckk = C[k][k].x;
        C[j][k] = cmake(v.x / ckk, v.y / ckk);
        dj -= C[j][k].x * C[j][k].x + C[j][k].y * C[j][k].y;
      }
      if (!(dj > 1.0e-13 * mjj) || !(mjj > 0.0)) break;          // dependent direction: the cycle ends here
      const double cjj = sqrt(dj);
      C[j][j] = cmake(cjj, 0.0);
      // t_j = ((W^H r_0)_j - sum_k C[j][k] t_k) / C[j][j],  (W^H r_0)_j = w_j^H r_0 = conj(G(0, j + 1))
      cplx tj = G(0, j + 1);
      tj = cmake(tj.x, -tj.y);
#pragma unroll
      for (int k = 0; k < SW_GRAM_MAXL; ++k) {
        if (k >= j) break;
        const cplx a_ = C[j][k], b_ = t[k];
        tj = cmake(tj.x - (a_.x * b_.x - a_.y * b_.y), tj.y - (a_.x * b_.y + a_.y * b_.x));
      }
      t[j] = cmake(tj.x / cjj, tj.y / cjj);
      res2 -= t[j].x * t[j].x + t[j].y * t[j].y;
      Leff = j + 1;
      rr = sqrt(fmax(res2, 0.0)) / nb_;
      if (s.iters[col] < 0 && rr < tol) s.iters[col] = iter_base + j + 1;
    }
  }
  // back substitution C^H y = t over the Leff directions that entered
  cplx y[SW_GRAM_MAXL];
#pragma unroll
  for (int j = SW_GRAM_MAXL - 1; j >= 0; --j) {
    y[j] = cmake(0.0, 0.0);
    if (j >= Leff) continue;
    cplx v = t[j];
#pragma unroll
    for (int k = SW_GRAM_MAXL - 1; k > 0; --k) {
      if (k <= j || k >= Leff) continue;
      // v -= conj(C[k][j]) y[k]
      const cplx a_ = C[k][j], b_ = y[k];
      v = cmake(v.x - (a_.x * b_.x + a_.y * b_.y), v.y - (a_.x * b_.y - a_.y * b_.x));
    }
    const double cjj = C[j][j].x;
    y[j] = cmake(v.x / cjj, v.y / cjj);
  }
#pragma unroll
  for (int j = 0; j < SW_GRAM_MAXL; ++j)
    if (j < L) s.ys[(size_t)j * nbp + col] = y[j];
  s.relres[col] = cmake(rr, 0.0);
  if (!done && !(rr < tol_stop)) atomicAdd(notconv, 1);
}

__device__ __forceinline__ void fg_tail_col(const FgTail& t, int col) {
  if (t.kind == SW_TAIL_GRAM) {
    fg_gram_col(t.s, t.j, t.h1, t.tol, t.tol_stop, t.iter_base, t.notconv, t.first_cycle, col);
    return;
  }
  if (t.kind == SW_TAIL_BEGIN) fg_begin_col(t.s, t.h1, t.first_cycle, col);
  else if (t.kind == SW_TAIL_HESS)
    fg_hess_col(t.s, t.j, t.h1, t.h2, t.nrm2, t.tol, t.tol_stop, t.iter_base, t.pyth, t.notconv, col);
  else if (t.kind == SW_TAIL_VERIFY) fg_verify_col(t.s, t.h1, t.tol, t.tol_stop, t.notconv, col);
}

// the same updates as a kernel of their own (fused_reduce off, or no reduction precedes them)
__global__ void k_fg_tail(FgTail t) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= t.s.nbp) return;
  fg_tail_col(t, col);
}

// y = H(0:k,0:k)^-1 g(0:k) per probe
__global__ void k_fg_solve(FgScalars s, int k) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= s.nbp) return;
  const int m = s.m, nbp = s.nbp;
  for (int i = k - 1; i >= 0; --i) {
    cplx t = s.g[(size_t)i * nbp + col];
    for (int l = i + 1; l < k; ++l) {
      const cplx hl = s.H[((size_t)i * m + l) * nbp + col];
      const cplx yl = s.y[(size_t)l * nbp + col];
      cfma(t, cmake(-hl.x, -hl.y), yl);
    }
    const cplx hii = s.H[((size_t)i * m + i) * nbp + col];
    const double dd = hii.x * hii.x + hii.y * hii.y;
    cplx yi = cmake(0.0, 0.0);
    if (dd > 0.0) yi = cmake((t.x * hii.x + t.y * hii.y) / dd, (t.y * hii.x - t.x * hii.y) / dd);
    s.y[(size_t)i * nbp + col] = yi;
    const double si = s.svec[(size_t)i * nbp + col].x;
    s.ys[(size_t)i * nbp + col] = cmake(si * yi.x, si * yi.y);
  }
}

// ------------------------------------------------------------------------------------------
// Completion of a cross-workgroup reduction INSIDE the launch that produced the partial sums
// ("last block done", two levels).  grid = (P row blocks, probe chunks); block p of a chunk has
// written partial[(p K + k) nbp + col] for its rows.  Blocks are grouped by SW_RED_GROUP consecutive p:
// the block that completes a group (ticket counter) adds the group's partials in the order of p into
// gpart[(g K + k) nbp + col]; the block that completes the last group adds the group sums in the order
// of g and writes out / coef, then applies `tail` to the chunk's 64 columns.  Which block does the
// adding depends on timing, the order of every sum does not: results are bit-reproducible, no
// floating-point atomics.  Tickets are zero on entry and left zero.  Two levels keep the data one CU has
// to pull small (a single finishing block would read P K KiB per chunk through one L1).
// Visibility across the chip's eight L2s WITHOUT cache-wide operations: a device-scope fence per
// workgroup (__threadfence = L2 write-back + invalidate) was measured at +25 us per reduction -- 512
// workgroups each walking their XCD's L2 --, 28.0k -> 21.2k probe-samples/s.  Instead the partial sums
// themselves are written and read with agent-scope relaxed atomic accesses (sc1: served at the
// device-coherent point, no non-coherent L2 copy), the writer drains its stores (s_waitcnt) before it
// takes the ticket, and the finishing block's reads depend on the ticket it drew.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_agent(cplx* p, cplx v) {
  __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ cplx ld_agent(const cplx* p) {
  cplx v;
  v.x = __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
// partial sum of a block: through the coherent path when the reduction is completed in this launch
__device__ __forceinline__ void st_partial(cplx* p, cplx v, bool coherent) {
  if (coherent) st_agent(p, v);
  else *p = v;
}
#define SW_RED_GROUP 32

struct RedArgs {
  int* tick1;           // [chunks][SW_RED_MAXGROUPS]
  int* tick2;           // [chunks]
  cplx* gpart;          // [groups][K][nbp]
  cplx* out;            // [K][nbp]
  const cplx* svec;     // optional, with coef: coef[k] = svec[k].x^2 out[k]
  cplx* coef;
};
#define SW_RED_MAXGROUPS 64

// sums `cnt` entries src[i * stride] (i < cnt) for KB values of k at a time: the four waves take
// i = wave, wave + 4, ... each, combine through LDS in wave order
// (red: KB * 4 * 64 values of LDS)
template <bool FINAL, int KB>
__device__ __forceinline__ void red_sum_block(const cplx* __restrict__ src, size_t istride, int cnt, int K,
                                              int nbp, size_t col, cplx* __restrict__ dst,
                                              const cplx* __restrict__ svec, cplx* __restrict__ coef,
                                              cplx* red) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int kb = 0; kb < K; kb += KB) {
    const int kn = min(KB, K - kb);
    for (int kk = 0; kk < kn; ++kk) {
      const cplx* sp = src + (size_t)(kb + kk) * nbp + col;
      cplx a0 = cmake(0.0, 0.0), a1 = a0;
      int i = wave;
      for (; i + 4 < cnt; i += 8) {
        const cplx u = ld_agent(sp + (size_t)i * istride);
        const cplx v = ld_agent(sp + (size_t)(i + 4) * istride);
        a0 = cadd(a0, u);
        a1 = cadd(a1, v);
      }
      if (i < cnt) a0 = cadd(a0, ld_agent(sp + (size_t)i * istride));
      red[(kk * 4 + wave) * 64 + lane] = cadd(a0, a1);
    }
    __syncthreads();
    for (int kk = wave; kk < kn; kk += 4) {
      const cplx* rk = red + (size_t)kk * 256 + lane;
      const cplx t = cadd(cadd(rk[0], rk[64]), cadd(rk[128], rk[192]));
      const size_t o = (size_t)(kb + kk) * nbp + col;
      if (FINAL) dst[o] = t;
      else st_agent(dst + o, t);         // a group sum: read by whichever block completes the chunk
      if (FINAL && coef) {
        const double q = svec[o].x;
        coef[o] = cmake(q * q * t.x, q * q * t.y);
      }
    }
    __syncthreads();
  }
}

// called by ALL threads of every block after the block's partials are stored (by any of its threads);
// red: KB * 256 values of LDS the caller no longer needs
template <int KB>
__device__ __forceinline__ void reduce_and_tail(const cplx* __restrict__ partial, int K, int nbp,
                                                const RedArgs& ra, const FgTail& tail, cplx* red) {
  __shared__ int s_flag;
  const int P = gridDim.x, p = blockIdx.x, chunk = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const size_t col = (size_t)chunk * 64 + lane;
  const int ngroups = (P + SW_RED_GROUP - 1) / SW_RED_GROUP;
  const int g = p / SW_RED_GROUP;
  const int g0 = g * SW_RED_GROUP;
  const int gcnt = min(SW_RED_GROUP, P - g0);
  // drain this wave's partial stores (sc1: acknowledged at the device-coherent point) before the
  // ticket; a workgroup-scope fence compiles to nothing here, hence the explicit wait
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int* t1 = ra.tick1 + chunk * SW_RED_MAXGROUPS + g;
    const int last = (atomicAdd(t1, 1) == gcnt - 1);
    if (last) atomicExch(t1, 0);
    s_flag = last;
  }
  __syncthreads();
  if (!s_flag) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const size_t pstride = (size_t)K * nbp;
  if (ngroups == 1) {
    red_sum_block<true, KB>(partial, pstride, gcnt, K, nbp, col, ra.out, ra.svec, ra.coef, red);
  } else {
    red_sum_block<false, KB>(partial + (size_t)g0 * pstride, pstride, gcnt, K, nbp, col,
                         ra.gpart + (size_t)g * pstride, nullptr, nullptr, red);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      int* t2 = ra.tick2 + chunk;
      const int last = (atomicAdd(t2, 1) == ngroups - 1);
      if (last) atomicExch(t2, 0);
      s_flag = last;
    }
    __syncthreads();
    if (!s_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    red_sum_block<true, KB>(ra.gpart, pstride, ngroups, K, nbp, col, ra.out, ra.svec, ra.coef, red);
  }
  if (tail.kind != SW_TAIL_NONE) {
    // (the sums were stored by this block's own threads: a block-level barrier orders them)
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 64) fg_tail_col(tail, (int)col);
  }
}

// ------------------------------------------------------------------------------------------
// Batched BLAS-1.  All of them: lane == probe, grid.y == 64-probe chunk.
// ------------------------------------------------------------------------------------------
#define SW_MAXK 34   // restart cap 32, + w itself + 1

template <class C>
struct PtrListT {
  const C* p[SW_MAXK];
};
typedef PtrListT<cplx> PtrList;

// partial[(blockIdx.x*K + k)*nbp + col] = sum over this block's rows of conj(V_k[r]) * W[r]
// CV: storage type of V and W (complex64 for the Krylov basis of the single-precision cycle mode);
// products and sums are fp64 either way
// fused (RedArgs::tick1 != NULL): the reduction is completed, and `tail` applied, inside this launch
template <int KT, class CV = cplx>
__global__ __launch_bounds__(SW_BLOCK) void k_multidot(PtrListT<CV> V, int K, const CV* __restrict__ W,
                                                       int n, int nbp, int rows_per_block,
                                                       cplx* __restrict__ partial, RedArgs ra, FgTail tail) {
  __shared__ cplx red[3][8][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  cplx acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = cmake(0.0, 0.0);
  // complex64 rows are half as wide: twice the rows in flight for the same bytes in flight
  constexpr int UNR = (sizeof(CV) == sizeof(cplxf)) ? 4 : 2;
#pragma unroll UNR
  for (int r = r0 + wave; r < r1; r += SW_WAVES_PER_BLOCK) {
    const size_t off = (size_t)r * nbp + col;
    const cplx w = widen(W[off]);
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) cfmac(acc[k], widen(V.p[k][off]), w);
  }
  // cross-wave reduction, 8 accumulators at a time
  for (int kb = 0; kb < KT; kb += 8) {
    if (kb >= K) break;
    if (wave > 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (kb + k < KT) red[wave - 1][k][lane] = acc[kb + k];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (kb + k < KT && kb + k < K) {
          cplx s = acc[kb + k];
          s = cadd(s, red[0][k][lane]);
          s = cadd(s, red[1][k][lane]);
          s = cadd(s, red[2][k][lane]);
          st_partial(&partial[((size_t)blockIdx.x * K + kb + k) * nbp + col], s, ra.tick1 != nullptr);
        }
      }
    }
    __syncthreads();
  }
  if (ra.tick1) reduce_and_tail<6>(partial, K, nbp, ra, tail, &red[0][0][0]);
}

// All inner products of NV vectors with each other in ONE pass:  G[a][b] = u_a^H u_b, a <= b  (K = NV (NV+1) / 2
// values, k = a NV - a (a - 1) / 2 + (b - a)); reduction completed in the launch, `tail` applied (fg_gram_col).
// Replaces the j + 1 inner-product and orthogonalisation passes per Arnoldi step of a restart cycle.
template <int NV>
__global__ __launch_bounds__(SW_BLOCK) void k_gram(PtrList U, int n, int nbp, int rows_per_block,
                                                   cplx* __restrict__ partial, RedArgs ra, FgTail tail) {
  constexpr int K = NV * (NV + 1) / 2;
  constexpr int KL = (K > 8) ? K : 8;          // (>= 6 * 4 * 64 values for the reduction tail)
  __shared__ cplx red[3][KL][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  cplx acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = cmake(0.0, 0.0);
#pragma unroll 2
  for (int r = r0 + wave; r < r1; r += SW_WAVES_PER_BLOCK) {
    const size_t off = (size_t)r * nbp + col;
    cplx u[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) u[a] = U.p[a][off];
    int k = 0;
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
      for (int b = a; b < NV; ++b) cfmac(acc[k++], u[a], u[b]);
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave - 1][k][lane] = acc[k];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      cplx v = acc[k];
      v = cadd(v, red[0][k][lane]);
      v = cadd(v, red[1][k][lane]);
      v = cadd(v, red[2][k][lane]);
      st_partial(&partial[((size_t)blockIdx.x * K + k) * nbp + col], v, ra.tick1 != nullptr);
    }
  }
  __syncthreads();
  if (ra.tick1) reduce_and_tail<6>(partial, K, nbp, ra, tail, &red[0][0][0]);
}

// out[k*nbp + col] = sum_p partial[(p*K + k)*nbp + col]
// grid = (K, nbp/64); the four waves of a block take p = wave, wave+4, ... with independent
// loads in flight, then combine through LDS in a fixed order (deterministic).
// optional second output (batched FGMRES on an UNNORMALISED basis, see k_fg_hess):
//   coef[k*nbp + col] = svec[k*nbp + col].x^2 * out[k*nbp + col]
__global__ __launch_bounds__(SW_BLOCK) void k_reduce_partials(const cplx* __restrict__ partial,
                                                              int P, int K, int nbp,
                                                              cplx* __restrict__ out,
                                                              const cplx* __restrict__ svec,
                                                              cplx* __restrict__ coef) {
  __shared__ cplx red[3][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const size_t pstride = (size_t)K * nbp;
  const cplx* src = partial + (size_t)k * nbp + col;
  cplx s0 = cmake(0.0, 0.0), s1 = s0, s2 = s0, s3 = s0;
  int p = wave;
  for (; p + 12 < P; p += 16) {
    const cplx a = src[(size_t)p * pstride];
    const cplx b = src[(size_t)(p + 4) * pstride];
    const cplx c = src[(size_t)(p + 8) * pstride];
    const cplx d = src[(size_t)(p + 12) * pstride];
    s0 = cadd(s0, a);
    s1 = cadd(s1, b);
    s2 = cadd(s2, c);
    s3 = cadd(s3, d);
  }
  for (; p < P; p += 4) s0 = cadd(s0, src[(size_t)p * pstride]);
  cplx s = cadd(cadd(s0, s1), cadd(s2, s3));
  if (wave > 0) red[wave - 1][lane] = s;
  __syncthreads();
  if (wave == 0) {
    s = cadd(s, red[0][lane]);
    s = cadd(s, red[1][lane]);
    s = cadd(s, red[2][lane]);
    out[(size_t)k * nbp + col] = s;
    if (svec) {
      const double q = svec[(size_t)k * nbp + col].x;
      coef[(size_t)k * nbp + col] = cmake(q * q * s.x, q * q * s.y);
    }
  }
}

// Wout[r] = Win[r] + sign * sum_k coef[k][col] * V_k[r]; optionally partial |Wout|^2 sums
// (as the real part of a cplx partial, layout as k_multidot with K = 1).
// CV: storage type of the V vectors (complex64 for the preconditioned directions Z of the
// single-precision preconditioner, widened on load); W32 (optional): complex64 copy of Wout, the
// next input of that preconditioner.
// CW: storage type of Win / Wout (complex64 when W is itself a vector of a complex64 Krylov basis).
template <int KT, bool NORM, class CV = cplx, class CW = cplx>
__global__ __launch_bounds__(SW_BLOCK) void k_multiaxpy(PtrListT<CV> V, int K,
                                                        const cplx* __restrict__ coef, double sign,
                                                        const CW* __restrict__ Win,
                                                        CW* __restrict__ Wout, int n, int nbp,
                                                        int rows_per_block,
                                                        cplx* __restrict__ partial,
                                                        cplxf* __restrict__ W32, RedArgs ra, FgTail tail) {
  __shared__ double redn[3][64];
  __shared__ cplx redt[NORM ? 256 : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  cplx c[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    c[k] = cmake(0.0, 0.0);
    if (k < K) {
      const cplx t = coef[(size_t)k * nbp + col];
      c[k] = cmake(sign * t.x, sign * t.y);
    }
  }
  double nrm = 0.0;
  constexpr int UNR = (sizeof(CV) == sizeof(cplxf) && sizeof(CW) == sizeof(cplxf)) ? 4 : 2;
#pragma unroll UNR
  for (int r = r0 + wave; r < r1; r += SW_WAVES_PER_BLOCK) {
    const size_t off = (size_t)r * nbp + col;
    cplx w = widen(Win[off]);
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) cfma(w, c[k], widen(V.p[k][off]));
    Wout[off] = narrow<CW>(w);
    if (W32) W32[off] = cmake((float)w.x, (float)w.y);
    if (NORM) nrm = fma(w.x, w.x, fma(w.y, w.y, nrm));
  }
  if (NORM) {
    if (wave > 0) redn[wave - 1][lane] = nrm;
    __syncthreads();
    if (wave == 0) {
      nrm += redn[0][lane];
      nrm += redn[1][lane];
      nrm += redn[2][lane];
      st_partial(&partial[(size_t)blockIdx.x * nbp + col], cmake(nrm, 0.0), ra.tick1 != nullptr);
    }
    if (ra.tick1) reduce_and_tail<1>(partial, 1, nbp, ra, tail, redt);
  }
}

// MR smoother update (one pass):  X += alpha*R ;  R -= alpha*T     alpha per probe
__global__ __launch_bounds__(SW_BLOCK) void k_mr_update(const cplx* __restrict__ alpha,
                                                        cplx* __restrict__ X, cplx* __restrict__ R,
                                                        const cplx* __restrict__ T, int n, int nbp,
                                                        int rows_per_block) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  const cplx al = alpha[col];
#pragma unroll 2
  for (int r = r0 + wave; r < r1; r += SW_WAVES_PER_BLOCK) {
    const size_t off = (size_t)r * nbp + col;
    cplx x = X[off], rr = R[off];
    const cplx t = T[off];
    cfma(x, al, rr);
    cfma(rr, cmake(-al.x, -al.y), t);
    X[off] = x;
    R[off] = rr;
  }
}

// dst = w * src   (w one complex constant: first Richardson step from a zero guess)
__global__ __launch_bounds__(SW_BLOCK) void k_cscale(cplx w, const cplx* __restrict__ src,
                                                     cplx* __restrict__ dst, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) dst[i] = cmul(w, src[i]);
}

// dst = a + b
__global__ __launch_bounds__(SW_BLOCK) void k_add(const cplx* __restrict__ a, const cplx* __restrict__ b,
                                                  cplx* __restrict__ dst, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) dst[i] = cadd(a[i], b[i]);
}

// dst[r] = src[srcrow[r]]  (row gather: the Pperm^T index shift in the internal row order)
__global__ __launch_bounds__(SW_BLOCK) void k_gather_rows(const int* __restrict__ srcrow,
                                                          const cplx* __restrict__ src,
                                                          cplx* __restrict__ dst, int n, int nbp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r = blockIdx.x * SW_WAVES_PER_BLOCK + wave;
  if (r >= n) return;
  const int s = srcrow ? srcrow[r] : r;
  dst[(size_t)r * nbp + col] = src[(size_t)s * nbp + col];
}

// ------------------------------------------------------------------------------------------
// Deflation (utils.py:221-225):  c = U^H x  (partials), then  out[r] = x[s] - sum_k U[s][k] c[k]
// with s = srcrow[r] (fused Pperm^T gather).  U is [n][kd] row-major, coefficients are
// wave-uniform so they ride the scalar path.
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(SW_BLOCK) void k_defl_dots(const cplx* __restrict__ U, int ldu,
                                                        int kd, const cplx* __restrict__ X, int n,
                                                        int nbp, int rows_per_block,
                                                        cplx* __restrict__ partial) {
  __shared__ cplx red[3][8][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  cplx acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = cmake(0.0, 0.0);
  for (int r = r0 + wave; r < r1; r += SW_WAVES_PER_BLOCK) {
    const int ru = __builtin_amdgcn_readfirstlane(r);
    const cplx x = X[(size_t)ru * nbp + col];
    const cplx* u = U + (size_t)ru * ldu;
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < kd) cfmac(acc[k], u[k], x);
  }
  for (int kb = 0; kb < KT; kb += 8) {
    if (kb >= kd) break;
    if (wave > 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (kb + k < KT) red[wave - 1][k][lane] = acc[kb + k];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (kb + k < KT && kb + k < kd) {
          cplx s = acc[kb + k];
          s = cadd(s, red[0][k][lane]);
          s = cadd(s, red[1][k][lane]);
          s = cadd(s, red[2][k][lane]);
          partial[((size_t)blockIdx.x * kd + kb + k) * nbp + col] = s;
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(SW_BLOCK) void k_defl_apply(const cplx* __restrict__ U, int kd,
                                                         const cplx* __restrict__ c,
                                                         const int* __restrict__ srcrow,
                                                         const cplx* __restrict__ X,
                                                         cplx* __restrict__ out, int n, int nbp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const int r = __builtin_amdgcn_readfirstlane(blockIdx.x * SW_WAVES_PER_BLOCK + wave);
  if (r >= n) return;
  const int s = srcrow ? srcrow[r] : r;
  cplx x = X[(size_t)s * nbp + col];
  const cplx* u = U + (size_t)s * kd;
#pragma unroll 4
  for (int k = 0; k < kd; ++k) {
    const cplx ck = c[(size_t)k * nbp + col];
    cfma(x, cmake(-u[k].x, -u[k].y), ck);
  }
  out[(size_t)r * nbp + col] = x;
}

// ------------------------------------------------------------------------------------------
// GPU-side multigrid setup (the device counterpart of multigrid.py:232-280:
// per-aggregate orthonormalisation of the test vectors, R = P^H, A_c = R A P).
// Test vectors of a level live in an ordinary level vector [n][nbp]: column = test vector.
// ------------------------------------------------------------------------------------------
#define SW_TV 8   // test vectors per chirality half (coarse sites carry 2 * SW_TV = 16 dofs = one MFMA tile)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave = one (aggregate, chirality) block of rpb rows x SW_TV columns: modified Gram-Schmidt,
// two sweeps ("twice is enough"), columns normalised.  rows[b*rpb + p] = level row of member p.
//   Q   [b][p][k]       (k fastest)  -> source of the prolongator values
//   Rv  = conj(Q)       grouped-ELL values of R = P^H with G = SW_TV, K = rpb (cols = rows[])
template <int RPL>   // rows per lane: rpb <= 64 * RPL
__global__ __launch_bounds__(SW_BLOCK) void k_block_qr(const cplx* __restrict__ V, int nbp,
                                                       const int* __restrict__ rows, int nblocks,
                                                       int rpb, cplx* __restrict__ Q,
                                                       cplx* __restrict__ Rv) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (b >= nblocks) return;
  cplx m[RPL][SW_TV];
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int p = lane + 64 * q;
#pragma unroll
    for (int k = 0; k < SW_TV; ++k) m[q][k] = cmake(0.0, 0.0);
    if (p < rpb) {
      const cplx* src = V + (size_t)rows[(size_t)b * rpb + p] * nbp;
#pragma unroll
      for (int k = 0; k < SW_TV; ++k) m[q][k] = src[k];
    }
  }
  for (int sweep = 0; sweep < 2; ++sweep) {
#pragma unroll
    for (int k = 0; k < SW_TV; ++k) {
      double n2 = 0.0;
#pragma unroll
      for (int q = 0; q < RPL; ++q) n2 += m[q][k].x * m[q][k].x + m[q][k].y * m[q][k].y;
      n2 = wave_sum(n2);
      const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
      for (int q = 0; q < RPL; ++q) m[q][k] = cmake(m[q][k].x * inv, m[q][k].y * inv);
#pragma unroll
      for (int j = k + 1; j < SW_TV; ++j) {
        cplx d = cmake(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < RPL; ++q) cfmac(d, m[q][k], m[q][j]);
        d.x = wave_sum(d.x);
        d.y = wave_sum(d.y);
#pragma unroll
        for (int q = 0; q < RPL; ++q) cfma(m[q][j], cmake(-d.x, -d.y), m[q][k]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int p = lane + 64 * q;
    if (p < rpb) {
      const size_t o = ((size_t)b * rpb + p) * SW_TV;
#pragma unroll
      for (int k = 0; k < SW_TV; ++k) {
        Q[o + k] = m[q][k];
        Rv[o + k] = cmake(m[q][k].x, -m[q][k].y);
      }
    }
  }
}

// vals[i] = map[i] >= 0 ? Q[map[i]] : 0   (prolongator values into their grouped-ELL slots)
__global__ __launch_bounds__(SW_BLOCK) void k_fill_from_map(const long long* __restrict__ map,
                                                            const cplx* __restrict__ Q,
                                                            cplx* __restrict__ vals, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    const long long s = map[i];
    vals[i] = s >= 0 ? Q[s] : cmake(0.0, 0.0);
  }
}

// Galerkin coarse operator by colour probing.  Coarse sites J = yc*Lc + xc carry 16 dofs; with the
// 16 colours q(J) = (xc & 3) + 4 (yc & 3) the five sites of any 5-point neighbourhood have distinct
// colours (Lc % 4 == 0), so the 256 columns col = 16 q + c of
//   E[(J, c')][col] = [c == c'] [q(J) == q],    Z = R A P E
// hold every block of A_c exactly once:  A_c[(I,i),(J,c)] = Z[(I,i)][16 q(J) + c]  for J ~ I.
__global__ __launch_bounds__(SW_BLOCK) void k_probe_unit(cplx* __restrict__ E, int Lc, int nbp) {
  // grid.x = coarse rows / 4, one wave per row; nbp == 256
  const int row = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= Lc * Lc * 16) return;
  const int lane = threadIdx.x & 63;
  const int J = row >> 4, cp = row & 15;
  const int q = ((J % Lc) & 3) + 4 * ((J / Lc) & 3);
  for (int col = lane; col < nbp; col += 64)
    E[(size_t)row * nbp + col] = cmake((col == 16 * q + cp) ? 1.0 : 0.0, 0.0);
}

// MFMA block-row form of A_c from Z: tile rt = coarse site I, k-step ks = 4 n + g for the n-th
// neighbour site S (sorted list nbr[I][0..4]) and its column group g:
//   vals[(rt*20 + ks)*64 + lane] = Z[(16 I + (lane & 15))][16 q(S) + 4 g + (lane >> 4)]
__global__ __launch_bounds__(SW_BLOCK) void k_bsr_from_probe(const cplx* __restrict__ Z, int nbp,
                                                             const int* __restrict__ nbr, int Lc,
                                                             cplx* __restrict__ vals,
                                                             int* __restrict__ kcol) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);   // (rt, ks)
  if (item >= Lc * Lc * 20) return;
  const int rt = item / 20, ks = item - rt * 20;
  const int S = nbr[rt * 5 + (ks >> 2)];
  const int q = ((S % Lc) & 3) + 4 * ((S / Lc) & 3);
  const int g = ks & 3;
  vals[(size_t)item * 64 + lane] =
      Z[((size_t)rt * 16 + (lane & 15)) * nbp + 16 * q + 4 * g + (lane >> 4)];
  if (lane == 0) kcol[item] = S * 16 + 4 * g;
}

// block-row operator -> dense row-major [n][n] (D zeroed by the caller); one wave per (rt, ks).
// colrank (optional): the operator's columns belong to a SUBSET of the level's 16-row site tiles, and
// colrank[site] is that site's tile index in the dense matrix (the even sites of a Schur complement)
__global__ __launch_bounds__(SW_BLOCK) void k_bsr_to_dense(const cplx* __restrict__ vals,
                                                           const int* __restrict__ kcol, int RT,
                                                           int KS, int n, cplx* __restrict__ D,
                                                           const int* __restrict__ colrank,
                                                           const int* __restrict__ unused) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= RT * KS) return;
  const int rt = item / KS;
  const cplx v = vals[(size_t)item * 64 + lane];
  int c = kcol[item];
  if (colrank) c = colrank[c >> 4] * 16 + (c & 15);
  if (v.x != 0.0 || v.y != 0.0)   // padding k-steps repeat column 0 with zero values
    D[((size_t)rt * 16 + (lane & 15)) * n + c + (lane >> 4)] = v;
}

// grouped-ELL operator -> dense row-major [n][n] (D zeroed by the caller; padding entries are zero-valued
// and skipped); one thread per (group, k, g)
__global__ __launch_bounds__(SW_BLOCK) void k_ell_to_dense(const int* __restrict__ cols, const cplx* __restrict__ vals,
                                                           int K, int G, int ngroups, int n,
                                                           cplx* __restrict__ D) {
  const size_t i = (size_t)blockIdx.x * SW_BLOCK + threadIdx.x;
  const size_t total = (size_t)ngroups * K * G;
  if (i >= total) return;
  const int g = (int)(i % G);
  const size_t gk = i / G;
  const int grp = (int)(gk / K);
  const cplx v = vals[i];
  if (v.x != 0.0 || v.y != 0.0) D[((size_t)grp * G + g) * n + cols[gk]] = v;
}

// dense row-major [n][n] -> MFMA block-row form with every 4-column group (KS = n/4).
// colsite (optional): dense column tile t stands for the level's site tile colsite[t]
__global__ __launch_bounds__(SW_BLOCK) void k_dense_to_bsr(const cplx* __restrict__ D, int n,
                                                           cplx* __restrict__ vals,
                                                           int* __restrict__ kcol,
                                                           const int* __restrict__ colsite) {
  const int lane = threadIdx.x & 63;
  const int KS = n >> 2;
  const size_t item = (size_t)blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (item >= (size_t)(n >> 4) * KS) return;
  const int rt = (int)(item / KS), ks = (int)(item - (size_t)rt * KS);
  vals[item * 64 + lane] = D[((size_t)rt * 16 + (lane & 15)) * n + 4 * ks + (lane >> 4)];
  if (lane == 0) kcol[item] = colsite ? colsite[ks >> 2] * 16 + 4 * (ks & 3) : 4 * ks;
}

// W[r][col] *= 1 / sqrt(nrm2[col].x)   (0 where the norm vanishes): normalisation of a basis vector
__global__ __launch_bounds__(SW_BLOCK) void k_colscale(cplx* __restrict__ W, const cplx* __restrict__ nrm2,
                                                       int n, int nbp) {
  const int lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.y * 64 + lane;
  const double q = nrm2[col].x;
  const double sc = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
  for (int r = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6); r < n; r += gridDim.x * SW_WAVES_PER_BLOCK) {
    const size_t off = (size_t)r * nbp + col;
    const cplx v = W[off];
    W[off] = cmake(sc * v.x, sc * v.y);
  }
}

// Y[rows of tile tmap[rt]] = X[same rows] for rt < RT (Y zeroed by the caller): restriction of a level
// vector to the site tiles a subset operator acts on
__global__ __launch_bounds__(SW_BLOCK) void k_copy_tiles(const cplx* __restrict__ X, const int* __restrict__ tmap,
                                                         int RT, int nbp, cplx* __restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);   // (rt, row in tile)
  if (item >= RT * 16) return;
  const size_t row = (size_t)tmap[item >> 4] * 16 + (item & 15);
  for (int c = lane; c < nbp; c += 64) Y[row * nbp + c] = X[row * nbp + c];
}

// deterministic pseudo-random start vectors (splitmix64 of (seed, row, column)), entries in [-1,1)^2
__global__ __launch_bounds__(SW_BLOCK) void k_fill_random(cplx* __restrict__ V, int n, int nbp,
                                                          int ncols, unsigned long long seed) {
  const int row = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n) return;
  const int col = threadIdx.x & 63;
  if (col >= nbp) return;
  unsigned long long z = seed + 0x9e3779b97f4a7c15ull * ((unsigned long long)row * 64ull + col + 1ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  const double re = (double)(z & 0xffffffffull) / 2147483648.0 - 1.0;
  const double im = (double)(z >> 32) / 2147483648.0 - 1.0;
  V[(size_t)row * nbp + col] = (col < ncols) ? cmake(re, im) : cmake(0.0, 0.0);
}

// MR step length alpha = <T,R>/<T,T> from d0 = <R,T> (= sum conj(R) T), d1 = <T,T>
__global__ void k_mr_alpha(const cplx* __restrict__ d, int nbp, cplx* __restrict__ alpha) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= nbp) return;
  const cplx d0 = d[col];
  const double d1 = d[(size_t)nbp + col].x;
  alpha[col] = (d1 > 0.0) ? cmake(d0.x / d1, -d0.y / d1) : cmake(0.0, 0.0);
}

// final estimates: e = a - b (b may be NULL)
__global__ void k_est_combine(const cplx* __restrict__ a, const cplx* __restrict__ b, int nbp,
                              cplx* __restrict__ e) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= nbp) return;
  e[col] = b ? csub(a[col], b[col]) : a[col];
}


// ------------------------------------------------------------------------------------------
// Block (subspace) algebra of the device eigensolver (sw_eig_*: the counterpart of ARPACK's
// eigs / eigsh at multigrid.py:174, utils.py:140): a block of up to 64 vectors is one [n][64]
// array (column = vector), so cross-vector products are cross-LANE products -- the one place
// outside the coarse operators where the fp64 matrix cores are the natural tool.
//   k_block_gram:   C[64][64] = V^H W          partial sums per row block, k_block_gram_reduce adds
//                   them in block order (deterministic)
//   k_block_rotate: out = W Y  or  out = C - W Y,  Y[64][64] (Ritz rotation, Cholesky-QR
//                   back-substitution, block residual)
//   k_row_sign:     dst = diag(sign) src       (gamma_3 in the level's row order)
// v_mfma_f64_16x16x4_f64 operand layout as for k_bsr_mfma: lane l holds A[l&15][l>>4],
// B[l>>4][l&15], D[(l>>4)+4r][l&15].  Here A = (V^H)[i][row], B = W[row][j]: both operands are plain
// loads from the [n][64] arrays (16 consecutive columns of 4 consecutive rows per wave).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SW_BLOCK) void k_block_gram(const cplx* __restrict__ V,
                                                         const cplx* __restrict__ W, int n,
                                                         int rows_per_block,
                                                         cplx* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int jb = threadIdx.x >> 6;       // column tile of W (wave)
  const int ib = blockIdx.y;             // column tile of V
  const int k = lane >> 4, c = lane & 15;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(n, r0 + rows_per_block);
  sw_double4 cr = {0.0, 0.0, 0.0, 0.0}, ci = {0.0, 0.0, 0.0, 0.0};
  const cplx* vp = V + (size_t)ib * 16 + c;
  const cplx* wp = W + (size_t)jb * 16 + c;
  for (int row = r0 + k; row < r1 + k; row += 4) {
    cplx v = cmake(0.0, 0.0), w = cmake(0.0, 0.0);
    if (row < r1) {
      v = vp[(size_t)row * 64];
      w = wp[(size_t)row * 64];
    }
    // conj(v) w = (vr wr + vi wi) + i (vr wi - vi wr)
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, w.x, cr, 0, 0, 0);
    cr = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, w.y, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, w.y, ci, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.y, w.x, ci, 0, 0, 0);
  }
  cplx* out = partial + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    out[(size_t)(ib * 16 + k + 4 * r) * 64 + jb * 16 + c] = cmake(cr[r], ci[r]);
}

__global__ __launch_bounds__(SW_BLOCK) void k_block_gram_reduce(const cplx* __restrict__ partial, int P,
                                                                cplx* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 4096) return;
  // four interleaved partial sums (a fixed tree: deterministic), so that four loads are in flight per thread
  cplx s0 = cmake(0.0, 0.0), s1 = s0, s2 = s0, s3 = s0;
  int p = 0;
  for (; p + 3 < P; p += 4) {
    s0 = cadd(s0, partial[(size_t)p * 4096 + e]);
    s1 = cadd(s1, partial[(size_t)(p + 1) * 4096 + e]);
    s2 = cadd(s2, partial[(size_t)(p + 2) * 4096 + e]);
    s3 = cadd(s3, partial[(size_t)(p + 3) * 4096 + e]);
  }
  for (; p < P; ++p) s0 = cadd(s0, partial[(size_t)p * 4096 + e]);
  out[e] = cadd(cadd(s0, s1), cadd(s2, s3));
}

// out[row][j] = sum_i W[row][i] Y[i][j]; Y in LDS (64 KB), one wave per row, lane = j
__global__ __launch_bounds__(SW_BLOCK) void k_block_rotate(const cplx* __restrict__ W,
                                                           const cplx* __restrict__ Y,
                                                           const cplx* __restrict__ Cm,
                                                           cplx* __restrict__ out, int n) {
  __shared__ cplx Ys[64 * 64];
  __shared__ cplx Wr[SW_WAVES_PER_BLOCK][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 4096; e += SW_BLOCK) Ys[e] = Y[e];
  __syncthreads();
  // n is a multiple of 4 on every level, so the four waves of a block run the same number of rounds
  for (int row = blockIdx.x * SW_WAVES_PER_BLOCK + wave; row < n; row += gridDim.x * SW_WAVES_PER_BLOCK) {
    Wr[wave][lane] = W[(size_t)row * 64 + lane];
    __syncthreads();
    cplx acc = cmake(0.0, 0.0);
#pragma unroll 8
    for (int i = 0; i < 64; ++i) cfma(acc, Wr[wave][i], Ys[i * 64 + lane]);
    // Cm given: out = Cm - W Y (the block residual W - V T of the Rayleigh-Ritz step)
    out[(size_t)row * 64 + lane] = Cm ? csub(Cm[(size_t)row * 64 + lane], acc) : acc;
    __syncthreads();
  }
}

__global__ __launch_bounds__(SW_BLOCK) void k_row_sign(const cplx* __restrict__ src,
                                                       const signed char* __restrict__ sign,
                                                       cplx* __restrict__ dst, int n, int nbp) {
  const int row = blockIdx.x * SW_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= n) return;
  const size_t o = (size_t)row * nbp + (size_t)blockIdx.y * 64 + (threadIdx.x & 63);
  const double sg = (double)sign[row];
  const cplx v = src[o];
  dst[o] = cmake(sg * v.x, sg * v.y);
}

}  // namespace swk
