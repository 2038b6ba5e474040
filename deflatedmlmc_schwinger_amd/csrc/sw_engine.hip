// sw_engine.hip -- host side of the MI355X Schwinger trace engine: operand ingestion, the
// batched flexible-GMRES / multigrid-cycle orchestration on one HIP stream, the probe-batch
// drivers and the C ABI of include/schwinger_hip.h.  gfx950 only; no CPU fallback: every
// compute entry point fails with a message when no HIP device is present.
#include "../../include/schwinger_hip.h"
#include "sw_kernels.hpp"
#include "sw_pack.hpp"
#include "sw_poly.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

using swk::cplx;
using swk::cplxf;
using swk::PtrList;

#define SW_MAXM 32  // engine restart cap (<= SW_MAX_KRYLOV)
#define SW_NC_SLOTS 2048
#define SW_TICKET_CHUNKS 128

static std::string g_create_error;
static int (*g_rccl_destroy)(void*) = nullptr;   // set when RCCL has been loaded (sw_comm_*)

struct sw_engine;
static int sw_fail(sw_engine* h, const char* fmt, ...);

#define HIPCHK(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return sw_fail(h, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                     __LINE__);                                                           \
  } while (0)
#define SWCHK(call)             \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != 0) return rc_;   \
  } while (0)

enum TimerCat { T_MVM = 0, T_DEFL, T_P, T_R, T_AXPY, T_DOTS, T_COARSEST, T_OTHER,
                T_STENCIL,      // k_stencil<0>  Y = A X
                T_STENCIL_RES,  // k_stencil<1>  Y = B - A X
                T_STENCIL_SM,   // k_stencil<2>  Y = X + w (B - A X)
                T_MFMA_DENSE,   // k_bsr_mfma on the dense coarsest inverse
                T_MFMA_OP,      // k_bsr_mfma on a block-structured level operator
                T_UNUSED13,     // (class 13 was a fused two-step stencil; removed, numbering kept)
                T_MFMA_OP2,     // k_bsr_mfma on level operators below level 1 of the solver hierarchy
                T_SCHUR,        // k_schur_step / k_eo_hop: even-odd smoother of the stencil level
                T_SCHUR_OP,     // k_schur_step<0/1>: operator / residual of the even-odd reduced system
                T_NCAT };
// classes >= T_STENCIL are folded into the mvm / coarsest buckets by sw_timers and reported
// separately by sw_kernel_stats

struct EllOp {
  int nrows = 0, ncols = 0, K = 0, G = 1, ngroups = 0;
  int* cols = nullptr;
  cplx* vals = nullptr;
  int* order = nullptr;     // processing order of the row groups (NULL: natural), sw_pack.hpp
  // prolongators of even-odd smoothed levels: the row groups of the EVEN sites only, in `order`'s
  // order -- the smoother never reads the odd half of the prolongated iterate (ensure_even_orders)
  int* order_even = nullptr;
  int ngroups_even = 0;
  bool set = false;
  // optional MFMA block-row form (square operators with n % 16 == 0 and dense-ish 16x4 blocks)
  int bsr_KS = 0;
  int* bsr_kcol = nullptr;   // [n/16][KS] first X row of each 4-column group
  cplx* bsr_vals = nullptr;  // [n/16][KS][64] lane-packed
  int* bsr_tmap = nullptr;   // optional: row tile -> tile of the output vector (subset operators)
  int bsr_RT = 0;            // row tiles when != nrows / 16 (subset operators)
  bool bsr_diag_last = false;   // the last four k-steps of every row tile are its own X rows (check_diag_last)
  // dense operator whose row tiles all list the same columns in the same order (kcol rows identical): the
  // LDS-staged dense kernel (k_dense_mfma3_lds) shares the X rows of a k-step between row tiles
  bool dense_uniform = false;
  // complex64 mirrors of the value arrays (same index arrays), made on demand for the
  // single-precision preconditioner (option precond_f32)
  cplxf* vals32 = nullptr;
  cplxf* bsr_vals32 = nullptr;
};

struct KrylovWS {
  int m = 0, n = 0, nbp = 0;
  cplx* V = nullptr;  // m vectors: vtilde_1 .. vtilde_m (vtilde_0 is the residual itself)
  cplx* Z = nullptr;  // m vectors
  // single-precision preconditioner on the lattice level: its m directions stay complex64 (the widening
  // is exact, so A z and x += Z y see the same numbers) and v32 is the complex64 copy of the Krylov
  // vector it is applied to next, written by the kernel that produced that vector
  cplxf* Z32 = nullptr;
  cplxf* v32 = nullptr;
  // option f32_krylov: the whole restart cycle in complex64 storage (basis V32, w = A z written
  // complex64, inner products accumulated in fp64) -- iterative refinement with GMRES(m) cycles: the
  // residual b - A x and the solution update stay fp64, once per restart
  cplxf* V32 = nullptr;
  cplx* xacc = nullptr;
  cplx* rres = nullptr;
  swk::FgScalars sc{};
  cplx* h1 = nullptr;   // [(m+2)][nbp]
  cplx* h2 = nullptr;   // [(m+2)][nbp]
  cplx* c1 = nullptr;   // [(m+2)][nbp] orthogonalisation coefficients svec_k^2 d_k
  cplx* nrm = nullptr;  // [nbp]
  cplx* gram = nullptr; // [15][nbp] inner products of a restart cycle in Gram-matrix form (fgmres_eo_gram)
};

struct Level {
  int n = 0;
  bool stencil = false;
  int L = 0;
  double mass = 0.0;
  cplx* U1 = nullptr;
  cplx* U2 = nullptr;
  EllOp A, P, R;
  EllOp Re;   // stencil level, even-odd reduced outer solve: R restricted to the even-site columns
  // small level solved DIRECTLY (sw_setup_level_inverse): dense inverse of the level operator in block-row
  // form; solve_dev applies it with one step of iterative refinement instead of iterating
  EllOp dinv;
  int nu_pre = 0, nu_post = 3, kcycle = 0;
  // fixed-polynomial (Richardson) smoother: weights 1/theta_k; empty -> adaptive MR steps
  std::vector<std::complex<double>> w_pre, w_post;
  bool rich = false;
  // even-odd post-smoother: Richardson weights for the Schur complement S of the even sites.
  // Stencil level: S is the fused two-hop kernel.  Coarse (block) levels: four subset operators in
  // MFMA block-row form, eo_op[0] = S (even x even, 9-point), [1] = F = A_eo D_oo^-1,
  // [2] = G = D_oo^-1, [3] = Hb = D_oo^-1 A_oe
  std::vector<std::complex<double>> w_eo;
  // the same smoother polynomial in product form (swp::product_form): x + q(S)(b' - S x) with
  // q(z) = (1 - prod_k (1 - w_k z)) / z = q_beta prod_j (1 - q_w[j] z); empty when not available
  std::vector<std::complex<double>> q_w;
  std::complex<double> q_beta;
  EllOp eo_op[5];   // [4] (optional): the DENSE inverse of S over the even sites' rows -- the level's Schur
                    // steps, and everything below the level, are then replaced by one application of it
  std::vector<int> h_rowmap;  // natural -> internal (empty: identity)
  int* rowmap = nullptr;
  // per-level cycle workspace, [n][nbp]
  int ws_nbp = 0;
  cplx *b = nullptr, *x = nullptr, *r = nullptr, *t = nullptr;
  // single-precision preconditioner: link mirrors, cycle workspace and the boundary pair
  // (i32: cast of the fp64 input, o32: result before the cast back)
  cplxf *U1f = nullptr, *U2f = nullptr;
  int ws32_nbp = 0;
  cplxf *b32 = nullptr, *x32 = nullptr, *r32 = nullptr, *t32 = nullptr, *i32 = nullptr, *o32 = nullptr;
  // reference-faithful smoother: gm_cycles restart cycles of unpreconditioned GMRES(gm_m) from a
  // zero guess, the engine's counterpart of lgmres(maxiter=smooth_iters) with inner_m = 30
  // (multigrid.py:393-394,438-439).  gm_m == 0: off
  int gm_m = 0, gm_cycles = 0;
  cplx *g1 = nullptr, *g2 = nullptr;
  KrylovWS gws;   // smoother workspace
  // GPU-side setup: test vectors of this level, [n][64] (column = test vector), and a swap buffer
  cplx *tv = nullptr, *tv2 = nullptr;
  KrylovWS kws;   // K-cycle workspace
  KrylovWS sws;   // outer-solve workspace
};

struct Hier {
  int nlevels = 0;
  Level lv[SW_MAX_LEVELS];
  EllOp cinv;
  bool ready = false;
  bool f32_valid = false;   // the complex64 mirrors match the operators (cleared by every setter)
  bool even_valid = false;  // the even-site group lists of the prolongators match (likewise)
};

struct EventRec {
  int cat;
  hipEvent_t e0, e1;
};

struct sw_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  // probe generation runs on a stream of its own, so that the next batch's probes are drawn while the
  // current batch is being solved (sw_probes_generate is asynchronous; sw_hutch_run waits for its slot)
  hipStream_t gen_stream = nullptr;
  int pb_slot = -1;
  std::string err;
  Hier hier[SW_MAX_HIER];
  int restart = 24;
  int solver_hid = 0;
  bool use_mfma = true;
  int bsr_stages = 4, dense_stages = 8;   // software-pipeline depth of k_bsr_mfma (k-steps in flight)
  int bsr_map = 1, bsr_sub = 8, dense_map = 0;   // block orderings of k_bsr_mfma (see the kernel)
  bool bsr_nt = true;     // non-temporal B loads / Y stores in k_bsr_mfma on level operators
  bool ell_order = true;   // visit prolongator row groups sorted by column (A/B switch)
  bool bsr_xreg = true;    // smoother step of a block operator: own X rows from the operand registers (A/B switch)
  bool p_even = true;      // prolongate onto the even sites only ahead of an even-odd smoother (A/B switch)
  int bench_what = 0;      // what sw_bench_dirac times: 0 operator, 1 restrict, 2 prolong, 3 coarsest inverse
  int bench_mode = 0;      // operator mode sw_bench_dirac times (0: Y=AX, 1: residual, 2: smoother step)
  bool mfma_ops = true;   // MFMA block-row kernel also for block-structured level operators
  bool mfma_3m = true;    // three real matrix products per complex one (k_bsr_mfma3) instead of four
  int mfma3_tiles = 0;    // tiles of 16 probes per wave in k_bsr_mfma3 (0: by the operator's size)
  int mfma_tiles = 4;
  int f32_tiles = 0;          // tiles of 16 probes per wave in k_bsr_mfma_f32 (0: automatic)
  int f32_stages = 4, f32_dense_stages = 8;
  int f32_splitk = 1;         // split-K block-row kernel: 0 off, 1 dense coarsest inverse, 2 every small operator
  bool f32_krylov = true;     // with precond_f32 on the lattice level: complex64 Krylov basis per restart cycle
  bool f32_pairs = true;      // two probes per lane in the HBM-bound complex64 kernels (A/B switch)
  int mfma_small_tiles = 2;   // tiles per wave for operators too small to fill the chip (0: off)
  // iteration count of the previous outer solve per (hierarchy, level): the convergence flag
  // is only read back from (hint - 2) on, earlier iterations are queued without a host sync
  int sync_hint[SW_MAX_HIER][SW_MAX_LEVELS] = {{0}};
  bool lazy_sync = true;
  // time-skewed order of the even-odd smoother's steps on the stencil level (schur_steps): -1 automatic
  // (strips sized for the Infinity Cache where the smoother's three half vectors exceed it), 0 off,
  // > 0 strip height in lattice rows
  int eo_skew = -1;
  bool eo_skew_chunk = false;   // walk the strips 64-probe chunk by chunk even where the full width is admissible
  // restart cycles of the even-odd reduced outer solve in Gram-matrix form (fgmres_eo_gram)
  bool gram_cycle = true;
  // even-odd smoother of the reduced-system cycle in product form (schur_product_steps): 2 nu + 2 half-vector
  // passes instead of 3 nu
  bool eo_product = true;
  int gj_block = 32;      // panel width of the blocked Gauss-Jordan inverse (0: unblocked)
  // fp64 even-odd kernels of the lattice level from LDS-staged halo tiles double-buffered by LDS-DMA, one
  // persistent workgroup per CU (k_schur_tile; 0 off, 4 / 8 waves per workgroup): the launches without a b'
  // operand (reduced operator, product-form factors) on full-lattice launches
  int eo_tile = 0;
  // dense operators (coarsest inverse, dense Schur inverse of a direct level, direct inverse of a small level)
  // through the LDS-staged three-product kernel k_dense_mfma3_lds (A/B switch)
  int dense_lds = 4;     // row tiles per workgroup (2 or 4); 0: k_bsr_mfma3.  2048^2 on 256 probes, per launch:
                         // k_bsr_mfma3 142 us, 2 tiles 157, 4 tiles 135 (profiles/r04_ab_sessions.txt, r04j)
  int eo_tile_dbg = 0;   // timing diagnostics of k_schur_tile (results are then wrong): 1 no prefetch, 2 no compute
  int num_cus = 256;
  // levels that carry the dense inverse of their operator (sw_setup_level_inverse) are solved with it
  bool direct_small = true;
  bool lgmres_aug = true;   // reference-faithful smoother: LGMRES's augmentation vector in the second cycle
  bool stencil_nt = false;
  int stencil_tile = 0;   // 0: automatic
  int stencil_spw = 0;    // 0: automatic (4)
  // one Gram-Schmidt pass per Arnoldi step instead of two; every outer solve is then verified
  // against its TRUE residual and continued when the recurrence was optimistic
  bool cgs2 = false;
  bool verify = true;
  // second Gram-Schmidt pass in the short inner Krylov cycles (K-cycle, GMRES smoother): they are
  // preconditioners of 2-30 steps whose result feeds a flexible outer iteration, one pass suffices
  bool inner_cgs2 = false;
  // last Arnoldi step of every restart cycle without its orthogonalisation pass: h_{j+1,j} from
  // |A z|^2 - sum |h_{k,j}|^2 (single-pass Gram-Schmidt only; see fgmres)
  bool pyth_last = true;
  // outer solves of an even-odd smoothed stencil level on the even-odd reduced (Schur complement)
  // system: half-length Krylov vectors, see fgmres_eo
  bool eo_solve = true;
  // block levels that carry the dense inverse of their Schur complement (eo_op[4]) are solved with it
  bool eo_direct = true;
  // single-precision preconditioner: every application of a multigrid cycle as the preconditioner of
  // an fp64 flexible GMRES (and sw_vcycle) runs in complex64 on the f32 matrix cores -- operands cast
  // at the boundary, residuals / orthogonalisation / verification stay fp64
  bool precond_f32 = false;
  // deflation
  int kd = 0;
  cplx* U = nullptr;  // [n0][kd] internal row order
  // MLMC-level deflation vectors V_l (utils.py:260-266), [n_l][k_l] internal row order
  int lkd[SW_MAX_LEVELS] = {0};
  cplx* lV[SW_MAX_LEVELS] = {nullptr};
  // Pperm^T gathers and MLMC rhs maps (hid 0)
  int* perm_src[SW_MAX_LEVELS] = {nullptr};
  EllOp rhsmap[SW_MAX_LEVELS];
  // scratch
  cplx* partial = nullptr;
  size_t partial_bytes = 0;
  cplx* small = nullptr;  // small per-probe scalars
  size_t small_bytes = 0;
  void* stage = nullptr;  // host<->device staging
  size_t stage_bytes = 0;
  std::vector<std::pair<void*, size_t>> allocs;
  // probe batch state
  int pb_level = -1, pb_nb = 0, pb_nbp = 0;
  int8_t* pb_probes = nullptr;   // currently selected slot
  struct ProbeSlot {
    int8_t* p = nullptr;
    size_t bytes = 0;
    int level = -1, nb = 0;
    hipEvent_t ready = nullptr;   // recorded on gen_stream behind the slot's generation (pending: true)
    bool pending = false;
  };
  std::vector<ProbeSlot> slots;
  // device probe stream (k_mt_jump / k_mt_generate): window = 624 raw MT19937 words at mt_pos
  bool mt_set = false;
  uint32_t* mt_win0 = nullptr;      // window at stream position 0 (as handed to sw_probes_stream_set)
  uint32_t* mt_win = nullptr;       // [2][624] double buffer, mt_win + 624*mt_cur is current
  int mt_cur = 0;
  uint64_t mt_pos = 0;              // stream position (draws) of the current window
  uint32_t* mt_poly = nullptr;      // [624] x^mt_dist mod phi
  uint64_t mt_dist = 0;
  uint32_t* mt_family = nullptr;    // [mt_fam_count][624]: x^(s*mt_fam_seg), s = 1..count
  uint64_t mt_fam_seg = 0;
  int mt_fam_count = 0;
  cplx *pb_x0 = nullptr, *pb_rhs = nullptr, *pb_z = nullptr, *pb_xc = nullptr, *pb_xc2 = nullptr,
       *pb_y = nullptr, *pb_w = nullptr, *pb_w2 = nullptr, *pb_xd = nullptr;
  int pb_ws_nbp = 0;
  cplx* pb_est = nullptr;
  int* pb_iters = nullptr;
  std::vector<int32_t> last_iters_f, last_iters_c;
  // profiling
  bool profiling = false;
  std::vector<EventRec> recs;
  std::vector<hipEvent_t> evpool;
  double tacc[T_NCAT] = {0};
  int64_t tcount[T_NCAT] = {0};
  double twork[T_NCAT] = {0};   // arithmetic issued per class while profiling (flops, MFMA classes)
  int64_t launches = 0;
  // convergence counters: one slot per FGMRES scalar update of an outer solve (no per-iteration reset);
  // the last slot is the sink of the inner solves, which nobody reads
  int* d_notconv = nullptr;  // [SW_NC_SLOTS]
  int nc_next = 0;
  int* h_notconv = nullptr;  // pinned
  // tickets of the in-launch reductions (swk::reduce_and_tail), zero between launches
  int* d_tick = nullptr;     // [SW_TICKET_CHUNKS * (SW_RED_MAXGROUPS + 1)]
  bool fused_reduce = true;
  // the batch is iterated until every probe's residual is below stop_factor * tol; iteration counts are
  // reported at tol (multigrid.py:347-366).  1: the reference's stopping point; 0.1: per-probe estimates
  // to 1e-10 relative even where the estimate cancels to a small number
  double stop_factor = 1.0;
  // directly solved levels (sw_setup_level_inverse): the measured ||b - A x|| / ||b|| of the last such solve
  // per right-hand side, and how often the refinement did not reach the tolerance and the iterative path
  // took over
  std::vector<double> direct_relres;
  int64_t direct_fallbacks = 0;
  // host time spent inside hipMalloc / hipFree, calls and bytes allocated since creation
  double alloc_s = 0.0, alloc_bytes = 0.0;
  int64_t alloc_calls = 0, pool_hits = 0;
  // device eigensolver (sw_eig_*): three [n][64] block buffers on one (hierarchy, level), the gamma_3 signs
  // in that level's row order, the partial sums of the block Gram kernel
  cplx* eig_buf[3] = {nullptr, nullptr, nullptr};
  signed char* eig_sign = nullptr;
  cplx* eig_small = nullptr;     // [64*64] reduced Gram matrix / rotation
  int eig_hid = -1, eig_level = -1, eig_n = 0;
  void* comm = nullptr;      // RCCL communicator (sw_comm_init), one rank per engine
  double* d_stats = nullptr; // [4] all-reduce buffer
};

static int sw_fail(sw_engine* h, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  else g_create_error = buf;
  return 1;
}

// ---------------------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------------------
// hipMalloc / hipFree with their host time accumulated (option "alloc_seconds" etc.: the setup log splits its
// phases into allocation time and the rest).  Large blocks are PARKED instead of freed and handed out again:
// the device setup allocates and frees multi-GB scratch per level (11.8 GB for one Arnoldi run on the 1024^2
// lattice), and in a process that has already moved ~100 GB through the allocator those calls were measured at
// 2.5-3.4 s per setup where a fresh process needs 0.02 s (profiles/r04_ab_sessions.txt, r04h: the cause of the
// driver-observed 2.4x slower 1024^2 setup of round 3).  The pool is per process (engines come and go), capped
// (SW_POOL_GB, default 128 of the 288 GB: evicting parked blocks is itself a slow hipFree; 0 disables; an
// allocation that fails with blocks parked returns them to the driver and retries), and a block is parked only
// after the freeing engine's streams have drained (hipFree's implicit synchronisation is what made reuse by
// another stream safe before).
namespace {
struct ParkedBlock {
  void* p;
  size_t bytes;
  int device;
};
std::mutex g_pool_mu;
std::vector<ParkedBlock> g_pool;
size_t g_pool_bytes = 0;
const size_t kPoolMinBlock = (size_t)32 << 20;
size_t pool_cap() {
  static const size_t cap = [] {
    const char* e = std::getenv("SW_POOL_GB");
    const double gb = e ? std::atof(e) : 128.0;
    return gb > 0.0 ? (size_t)(gb * 1073741824.0) : (size_t)0;
  }();
  return cap;
}
}  // namespace

static int dev_alloc(sw_engine* h, void** p, size_t bytes) {
  if (bytes == 0) bytes = 16;
  const auto t0 = std::chrono::steady_clock::now();
  *p = nullptr;
  if (bytes >= kPoolMinBlock && pool_cap() > 0) {
    // size classes of 1/8 octave (at most 12.5 % over the request), so that workspaces of similar shapes --
    // Krylov bases of m or m + 1 vectors, the Arnoldi basis, probing blocks -- meet in the pool
    size_t top = (size_t)1 << 25;
    while ((top << 1) <= bytes) top <<= 1;
    const size_t step = top >> 3;
    bytes = ((bytes + step - 1) / step) * step;
  }
  size_t got = bytes;
  if (bytes >= kPoolMinBlock && pool_cap() > 0) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i].device == h->device && g_pool[i].bytes >= bytes && g_pool[i].bytes <= 2 * bytes &&
          (best < 0 || g_pool[i].bytes < g_pool[best].bytes))
        best = (int)i;
    if (best >= 0) {
      *p = g_pool[best].p;
      got = g_pool[best].bytes;
      g_pool_bytes -= got;
      g_pool.erase(g_pool.begin() + best);
      h->pool_hits++;
    }
  }
  if (!*p && hipMalloc(p, bytes) != hipSuccess) {
    // out of device memory with blocks parked: give them back and try once more
    (void)hipGetLastError();
    std::vector<ParkedBlock> blocks;
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      blocks.swap(g_pool);
      g_pool_bytes = 0;
    }
    for (auto& b : blocks) (void)hipFree(b.p);
    *p = nullptr;
    HIPCHK(hipMalloc(p, bytes));
  }
  h->alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  h->alloc_calls++;
  h->alloc_bytes += (double)bytes;
  h->allocs.push_back({*p, got});
  return 0;
}
static int dev_free(sw_engine* h, void* p) {
  if (!p) return 0;
  size_t bytes = 0;
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (h->allocs[i].first == p) {
      bytes = h->allocs[i].second;
      h->allocs.erase(h->allocs.begin() + i);
      break;
    }
  const auto t0 = std::chrono::steady_clock::now();
  if (bytes >= kPoolMinBlock && pool_cap() > 0) {
    // nothing of this engine may still be using the block when another stream gets it
    if (h->stream) HIPCHK(hipStreamSynchronize(h->stream));
    if (h->gen_stream) HIPCHK(hipStreamSynchronize(h->gen_stream));
    std::vector<void*> evict;
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      g_pool.push_back({p, bytes, h->device});
      g_pool_bytes += bytes;
      while (g_pool_bytes > pool_cap() && !g_pool.empty()) {     // oldest first
        evict.push_back(g_pool.front().p);
        g_pool_bytes -= g_pool.front().bytes;
        g_pool.erase(g_pool.begin());
      }
    }
    for (void* q : evict) HIPCHK(hipFree(q));
  } else {
    HIPCHK(hipFree(p));
  }
  h->alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}
template <class T>
static int dev_realloc(sw_engine* h, T** p, size_t count) {
  if (*p) SWCHK(dev_free(h, *p));
  *p = nullptr;
  void* q = nullptr;
  SWCHK(dev_alloc(h, &q, count * sizeof(T)));
  *p = (T*)q;
  return 0;
}
template <class T>
static int upload(sw_engine* h, T** dst, const T* src, size_t count) {
  SWCHK(dev_realloc(h, dst, count));
  if (count) HIPCHK(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}
static int ensure_stage(sw_engine* h, size_t bytes) {
  if (h->stage_bytes >= bytes) return 0;
  if (h->stage) SWCHK(dev_free(h, h->stage));
  h->stage = nullptr;
  SWCHK(dev_alloc(h, &h->stage, bytes));
  h->stage_bytes = bytes;
  return 0;
}
static int ensure_partial(sw_engine* h, size_t bytes) {
  if (h->partial_bytes >= bytes) return 0;
  if (h->partial) SWCHK(dev_free(h, h->partial));
  h->partial = nullptr;
  void* q;
  SWCHK(dev_alloc(h, &q, bytes));
  h->partial = (cplx*)q;
  h->partial_bytes = bytes;
  return 0;
}
static inline int pad64(int nb) { return ((nb + 63) / 64) * 64; }

// ---------------------------------------------------------------------------------------------
// launch bookkeeping (HIP-event buckets mirror CustomTimer, utils.py:366-445)
// ---------------------------------------------------------------------------------------------
struct LaunchScope {
  sw_engine* h;
  int idx = -1;
  LaunchScope(sw_engine* h_, int cat) : h(h_) {
    h->launches++;
    if (h->profiling) {
      EventRec r;
      r.cat = cat;
      auto get = [&]() {
        hipEvent_t e;
        if (!h->evpool.empty()) {
          e = h->evpool.back();
          h->evpool.pop_back();
        } else {
          (void)hipEventCreate(&e);
        }
        return e;
      };
      r.e0 = get();
      r.e1 = get();
      (void)hipEventRecord(r.e0, h->stream);
      h->recs.push_back(r);
      idx = (int)h->recs.size() - 1;
    }
  }
  ~LaunchScope() {
    if (idx >= 0) (void)hipEventRecord(h->recs[idx].e1, h->stream);
  }
};
static void harvest_events(sw_engine* h) {
  for (auto& r : h->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      h->tacc[r.cat] += ms;
      h->tcount[r.cat] += 1;
    }
    h->evpool.push_back(r.e0);
    h->evpool.push_back(r.e1);
  }
  h->recs.clear();
}
static int stream_sync(sw_engine* h) {
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->profiling) harvest_events(h);
  return 0;
}
#define KLAUNCH_CHECK() HIPCHK(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// CSR -> grouped ELL / MFMA block-row form: packed on the host (sw_pack.hpp), uploaded here
// ---------------------------------------------------------------------------------------------
static int build_ell(sw_engine* h, EllOp& op, int nrows, int ncols, const int64_t* indptr,
                     const int32_t* indices, const std::complex<double>* data,
                     const std::vector<int>& rows_int, const std::vector<int>& colmap,
                     int forceG = 0, bool sort_by_column = false) {
  swp::EllHost e;
  std::string err;
  if (swp::ell_pack(e, err, nrows, ncols, indptr, indices, data, rows_int, colmap, forceG,
                    sort_by_column) != 0)
    return sw_fail(h, "%s", err.c_str());
  op.nrows = nrows;
  op.ncols = ncols;
  op.K = e.K;
  op.G = e.G;
  op.ngroups = e.ngroups;
  SWCHK(upload(h, &op.cols, e.cols.data(), e.cols.size()));
  SWCHK(upload(h, (std::complex<double>**)&op.vals, e.vals.data(), e.vals.size()));
  if (!e.order.empty()) SWCHK(upload(h, &op.order, e.order.data(), e.order.size()));
  op.set = true;
  return 0;
}

// MFMA block-row form of a square CSR operator in natural order (level >= 1 operators and the
// dense inverse).  Built only when at least `min_fill` of the packed 16x4 groups is non-zero.
static void check_diag_last(EllOp& op, const int32_t* kcol, const int32_t* tmap);
static int build_bsr(sw_engine* h, EllOp& op, int n, const int64_t* indptr, const int32_t* indices,
                     const std::complex<double>* data, double min_fill) {
  swp::BsrHost b;
  swp::bsr_pack(b, n, indptr, indices, data, min_fill);
  op.bsr_KS = 0;
  if (b.KS == 0) return 0;
  SWCHK(upload(h, &op.bsr_kcol, b.kcol.data(), b.kcol.size()));
  SWCHK(upload(h, (std::complex<double>**)&op.bsr_vals, b.vals.data(), b.vals.size()));
  op.bsr_KS = b.KS;
  if (op.nrows == 0) op.nrows = n;
  check_diag_last(op, b.kcol.data(), nullptr);
  return 0;
}

// Does every row tile end with its own diagonal block (k-steps KS-4 .. KS-1 = X rows 16 ot + 4 r)?
// The packers arrange it (sw_pack.hpp, hierarchy.block_rows_from_matrix, setup_gpu.level_geometry);
// the kernel's register shortcut is only taken where this check, on the index array itself, passes.
static void check_diag_last(EllOp& op, const int32_t* kcol, const int32_t* tmap) {
  op.bsr_diag_last = false;
  const int KS = op.bsr_KS;
  const int RT = op.bsr_RT > 0 ? op.bsr_RT : op.nrows / 16;
  if (KS < 4 || KS % 4 || RT <= 0 || !kcol) return;
  for (int rt = 0; rt < RT; ++rt) {
    const int ot = tmap ? tmap[rt] : rt;
    for (int r = 0; r < 4; ++r)
      if (kcol[(size_t)rt * KS + KS - 4 + r] != 16 * ot + 4 * r) return;
  }
  op.bsr_diag_last = true;
}

static int free_op(sw_engine* h, EllOp& op) {
  SWCHK(dev_free(h, op.cols));
  SWCHK(dev_free(h, op.vals));
  SWCHK(dev_free(h, op.order));
  SWCHK(dev_free(h, op.order_even));
  SWCHK(dev_free(h, op.bsr_kcol));
  SWCHK(dev_free(h, op.bsr_vals));
  SWCHK(dev_free(h, op.bsr_tmap));
  SWCHK(dev_free(h, op.vals32));
  SWCHK(dev_free(h, op.bsr_vals32));
  op = EllOp();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// kernel launch wrappers
// ---------------------------------------------------------------------------------------------
static int launch_bsr(sw_engine* h, const EllOp& op, int mode, const cplx* X, const cplx* B, cplx* Y,
                      int nbp, int cat, cplx w) {
  const int RT = op.bsr_RT > 0 ? op.bsr_RT : op.nrows / 16;
  // MFMA column tiles per wave (8 probes each).  Small operators (a 4096-row level on 256 probes is
  // 2048 waves at NT = 4, two per SIMD once) are latency-bound: fewer tiles per wave = more waves
  int NT = h->mfma_tiles;
  if (h->mfma_small_tiles > 0 && (long long)RT * ((2 * nbp) / (16 * NT)) < 4096)
    NT = h->mfma_small_tiles;
  const int bmap = (cat == T_COARSEST) ? h->dense_map : h->bsr_map, msub = h->bsr_sub;
  const int RBn = (RT + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, NCn = (2 * nbp) / (16 * NT);
  const int want = (cat == T_COARSEST) ? h->dense_stages : h->bsr_stages;
  const int stages = (want >= 8 && op.bsr_KS % 8 == 0) ? 8 : ((want >= 4 && op.bsr_KS % 4 == 0) ? 4 : 2);
  dim3 grid = (bmap == 0) ? dim3(RBn, NCn) : dim3(RBn * NCn);
  // level-1 operator of the solver hierarchy vs. the smaller operators below it (separate stats)
  const int n1 = h->hier[h->solver_hid].nlevels > 1 ? h->hier[h->solver_hid].lv[1].n : 0;
  const int cls = cat == T_COARSEST ? T_MFMA_DENSE
                                    : (cat == T_MVM ? (op.nrows == n1 || n1 == 0 ? T_MFMA_OP : T_MFMA_OP2)
                                                    : cat);
  LaunchScope ls(h, cls);
  // 16 rows x 4 columns x nbp probes x 8 flops per complex multiply-add, per (tile, k-step)
  if (h->profiling) h->twork[cls] += 512.0 * (double)RT * (double)op.bsr_KS * (double)nbp;
  const int dl_kb = 4;     // k-steps per LDS stage (8 measured 135.1 us against 136.4: barriers are not the bound)
  if (h->mfma_3m && h->dense_lds > 0 && cat == T_COARSEST && mode == 0 && op.dense_uniform &&
      RT % h->dense_lds == 0 && op.bsr_KS % (dl_kb * SW_DL_DEPTH) == 0 && op.bsr_KS >= 2 * dl_kb * SW_DL_DEPTH &&
      (nbp & 31) == 0 && X != Y) {
    // dense operator: operands shared through LDS (register-staged, double-buffered), one workgroup per
    // (16 dense_lds)-row x 32-probe block -- 1 KiB (0.75 KiB) per wave and k-step through the L2 -> CU path
    // instead of 2
#define DL_LAUNCH(RTW_, KB_)                                                                                \
  hipLaunchKernelGGL((swk::k_dense_mfma3_lds<RTW_, KB_>), dim3((RT / RTW_) * (nbp / 32)), dim3(128 * RTW_), 0, \
                     h->stream, (const cplx*)op.bsr_vals, (const int*)op.bsr_kcol, op.bsr_KS, RT, X, Y, nbp,  \
                     (const int*)op.bsr_tmap)
    if (h->dense_lds == 4) DL_LAUNCH(4, 4);
    else DL_LAUNCH(2, 4);
#undef DL_LAUNCH
    KLAUNCH_CHECK();
    return 0;
  }
  if (h->mfma_3m) {
    // three real products per complex one (k_bsr_mfma3): tiles of 16 probes per wave
    // Measured (gpurun_out r03d): one tile of 16 probes per wave wins wherever it was compared -- the
    // dense 2048^2 Schur inverse 138 us against 234 (two tiles) and 391 (four): more tiles mean more
    // stage registers (204 + 106 at two tiles and eight stages: one wave per SIMD), and the kernel lives on
    // two or three co-resident waves hiding each other's L1 round trips.  Larger level operators (lattices
    // beyond 128^2) keep two tiles per wave when they have the waves to spare.
    int NT3 = 1;
    if (cat != T_COARSEST && (long long)RT * (nbp / 32) >= 16384) NT3 = 2;
    if (h->mfma3_tiles == 1 || h->mfma3_tiles == 2 || h->mfma3_tiles == 4) NT3 = h->mfma3_tiles;
    while (NT3 > 1 && nbp % (16 * NT3)) NT3 >>= 1;
    const int NC3 = nbp / (16 * NT3);
    const dim3 grid3(RBn * NC3);
    const bool ntio3 = h->bsr_nt && cat != T_COARSEST;
    const int xr = (op.bsr_diag_last && h->bsr_xreg) ? 1 : 0;
    const int bm3 = (bmap == 0) ? 0 : bmap;
#define B3_LAUNCH(MD, NTT, NTB, SG)                                                              \
  hipLaunchKernelGGL((swk::k_bsr_mfma3<MD, NTT, NTB, SG>), grid3, dim3(SW_BLOCK), 0, h->stream,   \
                     (const cplx*)op.bsr_vals, (const int*)op.bsr_kcol, op.bsr_KS, RT, X, B, Y, nbp, \
                     w, bm3, msub, (const int*)op.bsr_tmap, xr)
#define B3_NT(MD, NTB, SG)                        \
  do {                                            \
    if (NT3 == 4) B3_LAUNCH(MD, 4, NTB, SG);      \
    else if (NT3 == 2) B3_LAUNCH(MD, 2, NTB, SG); \
    else B3_LAUNCH(MD, 1, NTB, SG);               \
  } while (0)
    if (!ntio3 && mode == 0 && want >= 16 && NT3 == 1 && op.bsr_KS % 16 == 0) B3_LAUNCH(0, 1, false, 16);
    else if (!ntio3 && mode == 0 && want >= 8) B3_NT(0, false, 8);
    else if (mode == 0) { if (ntio3) B3_NT(0, true, 4); else B3_NT(0, false, 4); }
    else if (mode == 1) { if (ntio3) B3_NT(1, true, 4); else B3_NT(1, false, 4); }
    else { if (ntio3) B3_NT(3, true, 4); else B3_NT(3, false, 4); }
#undef B3_NT
#undef B3_LAUNCH
    KLAUNCH_CHECK();
    return 0;
  }
  const double* Xr = (const double*)X;
  const double* Br = (const double*)B;
  double* Yr = (double*)Y;
#define BSR_LAUNCH_S(MD, NTT, NTB, SG)                                                           \
  hipLaunchKernelGGL((swk::k_bsr_mfma<MD, NTT, NTB, SG>), grid, dim3(SW_BLOCK), 0, h->stream,   \
                     (const cplx*)op.bsr_vals, (const int*)op.bsr_kcol, op.bsr_KS, RT, Xr, Br,   \
                     Yr, 2 * nbp, nbp, w, bmap, msub, (const int*)op.bsr_tmap,                  \
                     (op.bsr_diag_last && h->bsr_xreg) ? 1 : 0)
#define BSR_LAUNCH(MD, NTT)                                                                     \
  do {                                                                                          \
    const bool ntio_ = h->bsr_nt && cat != T_COARSEST;                                          \
    if (stages == 8) {                                                                          \
      if (ntio_) BSR_LAUNCH_S(MD, NTT, true, 8);                                                \
      else BSR_LAUNCH_S(MD, NTT, false, 8);                                                     \
    } else if (stages == 4) {                                                                   \
      if (ntio_) BSR_LAUNCH_S(MD, NTT, true, 4);                                                \
      else BSR_LAUNCH_S(MD, NTT, false, 4);                                                     \
    } else {                                                                                    \
      if (ntio_) BSR_LAUNCH_S(MD, NTT, true, 2);                                                \
      else BSR_LAUNCH_S(MD, NTT, false, 2);                                                     \
    }                                                                                           \
  } while (0)
  if (NT == 2) {
    if (mode == 0) BSR_LAUNCH(0, 2);
    else if (mode == 1) BSR_LAUNCH(1, 2);
    else BSR_LAUNCH(3, 2);
  } else {
    if (mode == 0) BSR_LAUNCH(0, 4);
    else if (mode == 1) BSR_LAUNCH(1, 4);
    else BSR_LAUNCH(3, 4);
  }
#undef BSR_LAUNCH
#undef BSR_LAUNCH_S
  KLAUNCH_CHECK();
  return 0;
}

// even_only: write the rows of the even sites only (prolongation before an even-odd smoother)
static int launch_ell(sw_engine* h, const EllOp& op, int mode, const cplx* X, const cplx* B,
                      cplx* Y, int nbp, int cat, cplx w = cplx{0.0, 0.0}, bool even_only = false) {
  if (!op.set) return sw_fail(h, "operator not set");
  if (op.bsr_KS > 0 && (mode == 0 || mode == 1 || mode == 3) && h->use_mfma &&
      (cat == T_COARSEST || h->mfma_ops))
    return launch_bsr(h, op, mode, X, B, Y, nbp, cat, w);
  if (!op.cols || !op.vals)
    return sw_fail(h, "operator exists in MFMA block-row form only (built on the device): it needs "
                      "use_mfma = mfma_ops = 1");
  const bool ev = even_only && op.order_even && h->p_even;
  const int ng = ev ? op.ngroups_even : op.ngroups;
  const int* ord = ev ? op.order_even : (h->ell_order ? op.order : nullptr);
  dim3 grid((ng + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, nbp / 64);
  LaunchScope ls(h, cat);
#define ELL_CASE(GG)                                                                            \
  case GG:                                                                                      \
    if (mode == 0)                                                                              \
      hipLaunchKernelGGL((swk::k_ell<GG, 0>), grid, dim3(SW_BLOCK), 0, h->stream, op.cols,      \
                         op.vals, op.K, ng, ord, X, B, Y, nbp, w);      \
    else if (mode == 1)                                                                         \
      hipLaunchKernelGGL((swk::k_ell<GG, 1>), grid, dim3(SW_BLOCK), 0, h->stream, op.cols,      \
                         op.vals, op.K, ng, ord, X, B, Y, nbp, w);      \
    else if (mode == 2)                                                                         \
      hipLaunchKernelGGL((swk::k_ell<GG, 2>), grid, dim3(SW_BLOCK), 0, h->stream, op.cols,      \
                         op.vals, op.K, ng, ord, X, B, Y, nbp, w);      \
    else                                                                                        \
      hipLaunchKernelGGL((swk::k_ell<GG, 3>), grid, dim3(SW_BLOCK), 0, h->stream, op.cols,      \
                         op.vals, op.K, ng, ord, X, B, Y, nbp, w);      \
    break;
  switch (op.G) {
    ELL_CASE(1)
    ELL_CASE(2)
    ELL_CASE(4)
    ELL_CASE(8)
    ELL_CASE(16)
    default:
      return sw_fail(h, "unsupported ELL group size %d", op.G);
  }
#undef ELL_CASE
  KLAUNCH_CHECK();
  return 0;
}

static int launch_bsr(sw_engine* h, const EllOp& op, int mode, const cplx* X, const cplx* B, cplx* Y,
                      int nbp, int cat, cplx w);
static int ensure_level_ws(sw_engine* h, Level& lv, int nbp);

// Y = coarsest_inv X : fp64 MFMA block-row kernel when the size allows, grouped-ELL otherwise.
// A coarsest level that carries the even-odd operators and the dense inverse of its Schur complement
// (sw_set_eo_operator 0..3 + sw_setup_direct_level) is solved in even-odd form instead -- the same exact
// solve at a quarter of the dense flops: b'_e = b_e - F b_o ; x_e = S^-1 b'_e ; x_o = G b_o - Hb x_e
// (X and Y must be different arrays; the level's residual buffer is the scratch for b').
static int apply_coarsest(sw_engine* h, Hier& H, const cplx* X, cplx* Y, int nbp) {
  Level& lc = H.lv[H.nlevels - 1];
  if (h->eo_direct && H.nlevels > 1 && lc.eo_op[4].set && lc.eo_op[1].set && lc.eo_op[2].set &&
      lc.eo_op[3].set && X != Y) {
    SWCHK(ensure_level_ws(h, lc, nbp));
    SWCHK(launch_bsr(h, lc.eo_op[1], 1, X, X, lc.r, nbp, T_MVM, cplx{0.0, 0.0}));
    SWCHK(launch_bsr(h, lc.eo_op[4], 0, lc.r, nullptr, Y, nbp, T_COARSEST, cplx{0.0, 0.0}));
    SWCHK(launch_bsr(h, lc.eo_op[2], 0, X, nullptr, Y, nbp, T_MVM, cplx{0.0, 0.0}));
    return launch_bsr(h, lc.eo_op[3], 1, Y, Y, Y, nbp, T_MVM, cplx{0.0, 0.0});
  }
  if (!H.cinv.set) return sw_fail(h, "coarsest inverse not set");
  return launch_ell(h, H.cinv, 0, X, nullptr, Y, nbp, T_COARSEST);
}

static int stencil_spw(sw_engine* h, int tile_w) {
  // consecutive x-sites per wave (must divide the tile width)
  // measured (profiles/r01_stencil_tiles.txt): one site per wave keeps the most independent
  // loads in flight and beats the register sliding window (SPW 2/4/8) at every lattice size
  int spw = h->stencil_spw > 0 ? h->stencil_spw : 1;
  while (spw > 1 && tile_w % spw) spw >>= 1;
  return spw;
}

// the Wilson stencil of the lattice level; CI = complex64: X is a direction stored by the
// single-precision preconditioner (mode 0 only), widened on load
template <class CI, class CO>
static int launch_stencil(sw_engine* h, Level& lv, int mode, const CI* X, const cplx* B, CO* Y,
                          int nbp, cplx w) {
  {
    swk::StencilArgs a;
    a.L = lv.L;
    a.Vh = lv.L * lv.L / 2;
    a.diag = 4.0 + lv.mass;
    a.U1 = lv.U1;
    a.U2 = lv.U2;
    a.nbp = nbp;
    a.w = w;
    a.nt_store = h->stencil_nt ? 1 : 0;
    // x-tile: whole rows while three of them (2 KiB per site and 64-probe chunk) fit well
    // inside a 4-MiB L2, otherwise 256-site (or 64-site) tiles: 3.6 -> 4.6 TB/s on 1024^2
    a.tile_w = lv.L;
    if (lv.L > 256) a.tile_w = (lv.L % 256 == 0) ? 256 : ((lv.L % 64 == 0) ? 64 : lv.L);
    if (h->stencil_tile > 0 && lv.L % h->stencil_tile == 0) a.tile_w = h->stencil_tile;
    const int spw = stencil_spw(h, a.tile_w);
    const int V = lv.L * lv.L;
    const int waves = V / spw;
    const int bpc = (waves + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
    const int nchunks = nbp / 64;
    LaunchScope ls(h, mode == 0 ? T_STENCIL : (mode == 1 ? T_STENCIL_RES : T_STENCIL_SM));
#define ST_LAUNCH(MD, SP)                                                                       \
  hipLaunchKernelGGL((swk::k_stencil<MD, SP, CI, CO>), dim3(bpc * nchunks), dim3(SW_BLOCK), 0, h->stream, \
                     X, B, Y, a, bpc)
#define ST_MODE(SP)                                                  \
  do {                                                               \
    if (mode == 0) ST_LAUNCH(0, SP);                                 \
    else if constexpr (std::is_same<CI, cplx>::value && std::is_same<CO, cplx>::value) { \
      if (mode == 1) ST_LAUNCH(1, SP);                               \
      else ST_LAUNCH(2, SP);                                         \
    } else return sw_fail(h, "internal: complex64 stencil input supports Y = A X only"); \
  } while (0)
    if (spw == 8) ST_MODE(8);
    else if (spw == 4) ST_MODE(4);
    else if (spw == 2) ST_MODE(2);
    else ST_MODE(1);
#undef ST_MODE
#undef ST_LAUNCH
    KLAUNCH_CHECK();
    return 0;
  }
}

// Y = A X (mode 0), Y = B - A X (mode 1) or Y = X + w (B - A X) (mode 2) at a level
static int apply_op(sw_engine* h, Level& lv, int mode, const cplx* X, const cplx* B, cplx* Y,
                    int nbp, cplx w = cplx{0.0, 0.0}) {
  if (lv.stencil) return launch_stencil<cplx, cplx>(h, lv, mode, X, B, Y, nbp, w);
  return launch_ell(h, lv.A, mode == 2 ? 3 : mode, X, B, Y, nbp, T_MVM, w);
}

// Workgroups of a reducing BLAS-1 launch (row blocks x probe chunks; option dot_blocks).  Round 1's cap of
// 128 row blocks left a 64-probe batch on a 1024^2 lattice with 512 waves for 2 M rows: 1024 blocks took
// its inner products from 61.7 to 20.7 ms and the fused-norm updates from 61.7 to 30.2 ms per batch
// (165 -> 214 probe-samples/s); beyond ~1500 the second reduction stage grows faster than the first shrinks.
// With four or more probe chunks 128 row blocks already make 512+ workgroups, and the second stage
// (k_reduce_partials: 5 us at 128 partials, 17 us at 256) would eat what the first gains.
static int g_dot_blocks = 1024, g_dot_pmax = 1024;
static void row_blocking(int n, int nbp, bool reduce, int* P, int* rpb) {
  const int nchunks = nbp / 64;
  int p;
  if (reduce) {
    p = std::max(8, std::min(g_dot_pmax, g_dot_blocks / std::max(1, nchunks)));
    if (nchunks >= 4 && g_dot_pmax == 1024) p = std::min(p, 128);
  } else {
    p = std::max(8, 4096 / std::max(1, nchunks));
  }
  int r = (n + p - 1) / p;
  if (r < SW_WAVES_PER_BLOCK) r = SW_WAVES_PER_BLOCK;
  *rpb = r;
  *P = (n + r - 1) / r;
}

// Reduction bookkeeping of the reducing BLAS-1 launches.  fused_reduce (default): the cross-workgroup
// sum is completed inside the launch (last block done, two levels: swk::reduce_and_tail) and an FGMRES
// scalar update (`tail`) may ride along; otherwise a second launch (k_reduce_partials) and, for a tail,
// a third (k_fg_tail).
static const swk::FgTail kNoTail = [] { swk::FgTail t{}; t.kind = SW_TAIL_NONE; return t; }();

static bool fused_ok(sw_engine* h, int P, int nbp) {
  return h->fused_reduce && h->d_tick && nbp / 64 <= SW_TICKET_CHUNKS &&
         (P + SW_RED_GROUP - 1) / SW_RED_GROUP <= SW_RED_MAXGROUPS;
}
static swk::RedArgs red_args(sw_engine* h, bool fused, int P, int K, int nbp, cplx* out, const cplx* svec,
                             cplx* coef) {
  swk::RedArgs ra{};
  if (fused) {
    ra.tick1 = h->d_tick;
    ra.tick2 = h->d_tick + SW_TICKET_CHUNKS * SW_RED_MAXGROUPS;
    ra.gpart = h->partial + (size_t)P * K * nbp;
    ra.out = out;
    ra.svec = svec;
    ra.coef = coef;
  }
  return ra;
}
static int launch_tail(sw_engine* h, const swk::FgTail& tail) {
  LaunchScope ls(h, T_OTHER);
  const int nbp = tail.s.nbp, tb = 256;
  hipLaunchKernelGGL(swk::k_fg_tail, dim3((nbp + tb - 1) / tb), dim3(tb), 0, h->stream, tail);
  KLAUNCH_CHECK();
  return 0;
}

// out[k][col] = sum_r conj(V_k[r]) W[r],  k < K
template <class CV>
static int multidot(sw_engine* h, const swk::PtrListT<CV>& V, int K, const CV* W, int n, int nbp,
                    cplx* out, const cplx* svec = nullptr, cplx* coef = nullptr,
                    const swk::FgTail* tail = nullptr) {
  if (K < 1 || K > SW_MAXM + 2) return sw_fail(h, "multidot: K=%d out of range", K);
  int P, rpb;
  row_blocking(n, nbp, true, &P, &rpb);
  const int ngroups = (P + SW_RED_GROUP - 1) / SW_RED_GROUP;
  SWCHK(ensure_partial(h, (size_t)(P + ngroups) * K * nbp * sizeof(cplx)));
  const bool fused = fused_ok(h, P, nbp);
  const swk::RedArgs ra = red_args(h, fused, P, K, nbp, out, svec, coef);
  const swk::FgTail tl = (fused && tail) ? *tail : kNoTail;
  dim3 grid(P, nbp / 64);
  {
    LaunchScope ls(h, T_DOTS);
#define MD_CASE(KT)                                                                             \
  hipLaunchKernelGGL((swk::k_multidot<KT, CV>), grid, dim3(SW_BLOCK), 0, h->stream, V, K, W, n, nbp, \
                     rpb, h->partial, ra, tl)
    if (K <= 2) MD_CASE(2);
    else if (K <= 4) MD_CASE(4);
    else if (K <= 8) MD_CASE(8);
    else if (K <= 16) MD_CASE(16);
    else if (K <= 24) MD_CASE(24);
    else MD_CASE(34);
#undef MD_CASE
    KLAUNCH_CHECK();
  }
  if (fused) return 0;
  {
    LaunchScope ls(h, T_DOTS);
    hipLaunchKernelGGL(swk::k_reduce_partials, dim3(K, nbp / 64), dim3(SW_BLOCK), 0, h->stream,
                       h->partial, P, K, nbp, out, svec, coef);
    KLAUNCH_CHECK();
  }
  if (tail) SWCHK(launch_tail(h, *tail));
  return 0;
}

// out[k][col] = u_a^H u_b for all pairs a <= b of NV vectors (k = a NV - a (a-1)/2 + (b-a)) in one pass;
// the reduction is completed in the launch and `tail` (the Gram-cycle update) applied
static int multigram(sw_engine* h, const PtrList& U, int NV, int n, int nbp, cplx* out, const swk::FgTail* tail) {
  if (NV < 2 || NV > SW_GRAM_MAXL + 1) return sw_fail(h, "multigram: %d vectors out of range", NV);
  const int K = NV * (NV + 1) / 2;
  int P, rpb;
  row_blocking(n, nbp, true, &P, &rpb);
  const int ngroups = (P + SW_RED_GROUP - 1) / SW_RED_GROUP;
  SWCHK(ensure_partial(h, (size_t)(P + ngroups) * K * nbp * sizeof(cplx)));
  if (!fused_ok(h, P, nbp)) return sw_fail(h, "internal: the Gram cycle needs the in-launch reductions");
  const swk::RedArgs ra = red_args(h, true, P, K, nbp, out, nullptr, nullptr);
  const swk::FgTail tl = tail ? *tail : kNoTail;
  dim3 grid(P, nbp / 64);
  LaunchScope ls(h, T_DOTS);
#define GR_CASE(NN) hipLaunchKernelGGL((swk::k_gram<NN>), grid, dim3(SW_BLOCK), 0, h->stream, U, n, nbp, rpb, \
                                       h->partial, ra, tl)
  if (NV == 2) GR_CASE(2);
  else if (NV == 3) GR_CASE(3);
  else if (NV == 4) GR_CASE(4);
  else GR_CASE(5);
#undef GR_CASE
  KLAUNCH_CHECK();
  return 0;
}

// Wout = Win + sign * sum_k coef[k] V_k ; optional nrm[col].x = ||Wout||^2
template <class CV, class CW>
static int multiaxpy(sw_engine* h, const swk::PtrListT<CV>& V, int K, const cplx* coef, double sign,
                     const CW* Win, CW* Wout, int n, int nbp, cplx* nrm_out,
                     cplxf* w32 = nullptr, const swk::FgTail* tail = nullptr) {
  if (K < 1 || K > SW_MAXM + 2) return sw_fail(h, "multiaxpy: K=%d out of range", K);
  if (tail && !nrm_out) return sw_fail(h, "internal: a reduction tail needs the fused norm");
  int P, rpb;
  row_blocking(n, nbp, nrm_out != nullptr, &P, &rpb);
  const int ngroups = (P + SW_RED_GROUP - 1) / SW_RED_GROUP;
  if (nrm_out) SWCHK(ensure_partial(h, (size_t)(P + ngroups) * nbp * sizeof(cplx)));
  const bool fused = nrm_out && fused_ok(h, P, nbp);
  const swk::RedArgs ra = red_args(h, fused, P, 1, nbp, nrm_out, nullptr, nullptr);
  const swk::FgTail tl = (fused && tail) ? *tail : kNoTail;
  dim3 grid(P, nbp / 64);
  {
    LaunchScope ls(h, T_AXPY);
#define MA_CASE(KT)                                                                              \
  do {                                                                                           \
    if (nrm_out)                                                                                 \
      hipLaunchKernelGGL((swk::k_multiaxpy<KT, true, CV, CW>), grid, dim3(SW_BLOCK), 0, h->stream, V, K, \
                         coef, sign, Win, Wout, n, nbp, rpb, h->partial, w32, ra, tl);           \
    else                                                                                         \
      hipLaunchKernelGGL((swk::k_multiaxpy<KT, false, CV, CW>), grid, dim3(SW_BLOCK), 0, h->stream, V, \
                         K, coef, sign, Win, Wout, n, nbp, rpb, h->partial, w32, ra, tl);        \
  } while (0)
    if (K <= 2) MA_CASE(2);
    else if (K <= 4) MA_CASE(4);
    else if (K <= 8) MA_CASE(8);
    else if (K <= 16) MA_CASE(16);
    else if (K <= 24) MA_CASE(24);
    else MA_CASE(34);
#undef MA_CASE
    KLAUNCH_CHECK();
  }
  if (nrm_out && !fused) {
    LaunchScope ls(h, T_DOTS);
    hipLaunchKernelGGL(swk::k_reduce_partials, dim3(1, nbp / 64), dim3(SW_BLOCK), 0, h->stream,
                       h->partial, P, 1, nbp, nrm_out, (const cplx*)nullptr, (cplx*)nullptr);
    KLAUNCH_CHECK();
  }
  if (tail && !fused) SWCHK(launch_tail(h, *tail));
  return 0;
}

static int zero_vec(sw_engine* h, cplx* p, int n, int nbp) {
  LaunchScope ls(h, T_AXPY);
  HIPCHK(hipMemsetAsync(p, 0, (size_t)n * nbp * sizeof(cplx), h->stream));
  return 0;
}
static int copy_vec(sw_engine* h, cplx* dst, const cplx* src, int n, int nbp) {
  LaunchScope ls(h, T_AXPY);
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * nbp * sizeof(cplx), hipMemcpyDeviceToDevice,
                        h->stream));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// workspaces
// ---------------------------------------------------------------------------------------------
static int ensure_small(sw_engine* h, int nbp) {
  const size_t need = (size_t)(SW_MAX_DEFL + 8) * nbp * sizeof(cplx);
  if (h->small_bytes >= need) return 0;
  if (h->small) SWCHK(dev_free(h, h->small));
  h->small = nullptr;
  void* q;
  SWCHK(dev_alloc(h, &q, need));
  h->small = (cplx*)q;
  h->small_bytes = need;
  return 0;
}

static int ensure_level_ws(sw_engine* h, Level& lv, int nbp) {
  if (lv.ws_nbp == nbp && lv.b) return 0;
  const size_t cnt = (size_t)lv.n * nbp;
  SWCHK(dev_realloc(h, &lv.b, cnt));
  SWCHK(dev_realloc(h, &lv.x, cnt));
  SWCHK(dev_realloc(h, &lv.r, cnt));
  SWCHK(dev_realloc(h, &lv.t, cnt));
  lv.ws_nbp = nbp;
  return 0;
}

static int free_krylov(sw_engine* h, KrylovWS& w) {
  SWCHK(dev_free(h, w.V)); w.V = nullptr;
  SWCHK(dev_free(h, w.Z)); w.Z = nullptr;
  SWCHK(dev_free(h, w.Z32)); w.Z32 = nullptr;
  SWCHK(dev_free(h, w.v32)); w.v32 = nullptr;
  SWCHK(dev_free(h, w.V32)); w.V32 = nullptr;
  SWCHK(dev_free(h, w.xacc)); w.xacc = nullptr;
  SWCHK(dev_free(h, w.rres)); w.rres = nullptr;
  SWCHK(dev_free(h, w.sc.H)); w.sc.H = nullptr;
  SWCHK(dev_free(h, w.sc.cs)); w.sc.cs = nullptr;
  SWCHK(dev_free(h, w.sc.sn)); w.sc.sn = nullptr;
  SWCHK(dev_free(h, w.sc.g)); w.sc.g = nullptr;
  SWCHK(dev_free(h, w.sc.y)); w.sc.y = nullptr;
  SWCHK(dev_free(h, w.sc.normb)); w.sc.normb = nullptr;
  SWCHK(dev_free(h, w.sc.relres)); w.sc.relres = nullptr;
  SWCHK(dev_free(h, w.sc.svec)); w.sc.svec = nullptr;
  SWCHK(dev_free(h, w.sc.ys)); w.sc.ys = nullptr;
  SWCHK(dev_free(h, w.c1)); w.c1 = nullptr;
  SWCHK(dev_free(h, w.sc.iters)); w.sc.iters = nullptr;
  SWCHK(dev_free(h, w.h1)); w.h1 = nullptr;
  SWCHK(dev_free(h, w.h2)); w.h2 = nullptr;
  SWCHK(dev_free(h, w.nrm)); w.nrm = nullptr;
  SWCHK(dev_free(h, w.gram)); w.gram = nullptr;
  w.m = w.n = w.nbp = 0;
  return 0;
}

static int ensure_krylov(sw_engine* h, KrylovWS& w, int m, int n, int nbp, bool outer) {
  if (w.m == m && w.n == n && w.nbp == nbp && w.V) return 0;
  SWCHK(free_krylov(h, w));
  const size_t vec = (size_t)n * nbp;
  SWCHK(dev_realloc(h, &w.V, vec * m));
  SWCHK(dev_realloc(h, &w.Z, vec * m));
  if (outer) {
    SWCHK(dev_realloc(h, &w.xacc, vec));
    SWCHK(dev_realloc(h, &w.rres, vec));
  }
  SWCHK(dev_realloc(h, &w.sc.H, (size_t)(m + 1) * m * nbp));
  SWCHK(dev_realloc(h, &w.sc.cs, (size_t)m * nbp));
  SWCHK(dev_realloc(h, &w.sc.sn, (size_t)m * nbp));
  SWCHK(dev_realloc(h, &w.sc.g, (size_t)(m + 1) * nbp));
  SWCHK(dev_realloc(h, &w.sc.y, (size_t)m * nbp));
  SWCHK(dev_realloc(h, &w.sc.normb, (size_t)nbp));
  SWCHK(dev_realloc(h, &w.sc.relres, (size_t)nbp));
  SWCHK(dev_realloc(h, &w.sc.svec, (size_t)(m + 1) * nbp));
  SWCHK(dev_realloc(h, &w.sc.ys, (size_t)m * nbp));
  SWCHK(dev_realloc(h, &w.c1, (size_t)(m + 2) * nbp));
  SWCHK(dev_realloc(h, &w.sc.iters, (size_t)nbp));
  SWCHK(dev_realloc(h, &w.h1, (size_t)(m + 2) * nbp));
  SWCHK(dev_realloc(h, &w.h2, (size_t)(m + 2) * nbp));
  SWCHK(dev_realloc(h, &w.nrm, (size_t)nbp));
  SWCHK(dev_realloc(h, &w.gram, (size_t)((SW_GRAM_MAXL + 1) * (SW_GRAM_MAXL + 2) / 2) * nbp));
  w.sc.m = m;
  w.sc.nbp = nbp;
  w.m = m;
  w.n = n;
  w.nbp = nbp;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// the multigrid cycle (MG.one_mg_step, multigrid.py:369-447) with MR(nu) smoothing
// ---------------------------------------------------------------------------------------------
static int fgmres(sw_engine* h, Hier& H, int level, const cplx* B, cplx* X, double tol, int maxiter,
                  int m, bool outer, KrylovWS& ws, int nbp, int* iters_total,
                  bool use_precond = true, const cplx* aug = nullptr);

// nu MR steps on (X, R) with R = B - A X maintained
static int mr_smooth(sw_engine* h, Level& lv, cplx* X, cplx* R, int nu, int nbp) {
  SWCHK(ensure_small(h, nbp));
  cplx* d = h->small;                // [2][nbp]
  cplx* alpha = h->small + 2 * nbp;  // [nbp]
  for (int s = 0; s < nu; ++s) {
    SWCHK(apply_op(h, lv, 0, R, nullptr, lv.t, nbp));
    PtrList pl;
    pl.p[0] = R;
    pl.p[1] = lv.t;
    SWCHK(multidot(h, pl, 2, lv.t, lv.n, nbp, d));
    {
      LaunchScope ls(h, T_DOTS);
      hipLaunchKernelGGL(swk::k_mr_alpha, dim3((nbp + 255) / 256), dim3(256), 0, h->stream, d, nbp,
                         alpha);
      KLAUNCH_CHECK();
    }
    int P, rpb;
    row_blocking(lv.n, nbp, false, &P, &rpb);
    {
      LaunchScope ls(h, T_AXPY);
      hipLaunchKernelGGL(swk::k_mr_update, dim3(P, nbp / 64), dim3(SW_BLOCK), 0, h->stream, alpha,
                         X, R, lv.t, lv.n, nbp, rpb);
      KLAUNCH_CHECK();
    }
  }
  return 0;
}

// n fixed-weight Richardson steps  x <- x + w_k (B - A x), ping-ponging between `cur`
// (holding x on entry, ignored when from_zero) and `other`; *result = buffer with the answer
static int rich_steps(sw_engine* h, Level& lv, const cplx* Bin, cplx* cur, cplx* other,
                      const std::vector<std::complex<double>>& w, bool from_zero, int nbp,
                      cplx** result) {
  size_t k = 0;
  if (from_zero && !w.empty()) {
    LaunchScope ls(h, T_AXPY);
    const size_t count = (size_t)lv.n * nbp;
    hipLaunchKernelGGL(swk::k_cscale, dim3(2048), dim3(SW_BLOCK), 0, h->stream,
                       cplx{w[0].real(), w[0].imag()}, Bin, cur, count);
    KLAUNCH_CHECK();
    k = 1;
  }
  for (; k < w.size(); ++k) {
    SWCHK(apply_op(h, lv, 2, cur, Bin, other, nbp, cplx{w[k].real(), w[k].imag()}));
    std::swap(cur, other);
  }
  *result = cur;
  return 0;
}

static int vcycle_rich(sw_engine* h, Hier& H, int l, const cplx* Bin, cplx* Xout, int nbp);

static int vec_add(sw_engine* h, const cplx* a, const cplx* b, cplx* dst, int n, int nbp) {
  LaunchScope ls(h, T_AXPY);
  hipLaunchKernelGGL(swk::k_add, dim3(2048), dim3(SW_BLOCK), 0, h->stream, a, b, dst,
                     (size_t)n * nbp);
  KLAUNCH_CHECK();
  return 0;
}

// E = gm_cycles x GMRES(gm_m) applied to A_l e = R from a zero guess (unpreconditioned).  With two cycles
// and option lgmres_aug (default) the second cycle is LGMRES's: its search space is augmented by the
// first cycle's correction, as SciPy's lgmres(maxiter=2, inner_m=30, outer_k=3) does at
// multigrid.py:393-394,438-439 (the first outer iteration has nothing to augment with).
static int gmres_smooth(sw_engine* h, Hier& H, int l, const cplx* R, cplx* E, int nbp) {
  Level& lv = H.lv[l];
  const int m = lv.gm_m;
  const bool lg = h->lgmres_aug && lv.gm_cycles == 2 && m + 1 <= SW_MAXM;
  SWCHK(ensure_krylov(h, lv.gws, lg ? m + 1 : m, lv.n, nbp, false));
  SWCHK(fgmres(h, H, l, R, E, 0.0, m, m, false, lv.gws, nbp, nullptr, false));
  for (int c = 1; c < lv.gm_cycles; ++c) {
    SWCHK(apply_op(h, lv, 1, E, R, lv.g1, nbp));                       // g1 = R - A E
    SWCHK(fgmres(h, H, l, lv.g1, lv.g2, 0.0, m, m, false, lv.gws, nbp, nullptr, false, lg ? E : nullptr));
    SWCHK(vec_add(h, E, lv.g2, E, lv.n, nbp));
  }
  return 0;
}

static int vcycle(sw_engine* h, Hier& H, int l, const cplx* Bin, cplx* Xout, int nbp);

// MG.one_mg_step as written (multigrid.py:369-447): smooth, residual, restrict, recurse, prolong,
// residual, smooth -- with the GMRES(m) x cycles smoother above in place of lgmres(maxiter=2)
static int vcycle_gmres(sw_engine* h, Hier& H, int l, const cplx* Bin, cplx* Xout, int nbp) {
  Level& lv = H.lv[l];
  Level& lc = H.lv[l + 1];
  SWCHK(ensure_level_ws(h, lv, nbp));
  SWCHK(ensure_level_ws(h, lc, nbp));
  if (!lv.P.set || !lv.R.set) return sw_fail(h, "transfer operators of level %d not set", l);
  if (!lv.g1 || lv.gws.nbp != nbp) {
    SWCHK(dev_realloc(h, &lv.g1, (size_t)lv.n * nbp));
    SWCHK(dev_realloc(h, &lv.g2, (size_t)lv.n * nbp));
  }
  SWCHK(gmres_smooth(h, H, l, Bin, Xout, nbp));                         // :388-399  (x = 0 + e)
  SWCHK(apply_op(h, lv, 1, Xout, Bin, lv.r, nbp));                      // :402
  SWCHK(launch_ell(h, lv.R, 0, lv.r, nullptr, lc.b, nbp, T_R));         // :406
  SWCHK(vcycle(h, H, l + 1, lc.b, lc.x, nbp));                          // :413-416 at the coarsest
  SWCHK(launch_ell(h, lv.P, 2, lc.x, Xout, Xout, nbp, T_P));            // :429
  SWCHK(apply_op(h, lv, 1, Xout, Bin, lv.r, nbp));                      // :433
  SWCHK(gmres_smooth(h, H, l, lv.r, lv.t, nbp));                        // :438
  return vec_add(h, Xout, lv.t, Xout, lv.n, nbp);                       // :444
}

static int vcycle(sw_engine* h, Hier& H, int l, const cplx* Bin, cplx* Xout, int nbp) {
  if (l < H.nlevels - 1 && H.lv[l].gm_m > 0) return vcycle_gmres(h, H, l, Bin, Xout, nbp);
  if (l < H.nlevels - 1 && H.lv[l].rich) return vcycle_rich(h, H, l, Bin, Xout, nbp);
  const int last = H.nlevels - 1;
  if (l == last) {
    return apply_coarsest(h, H, Bin, Xout, nbp);
  }
  Level& lv = H.lv[l];
  Level& lc = H.lv[l + 1];
  SWCHK(ensure_level_ws(h, lv, nbp));
  SWCHK(ensure_level_ws(h, lc, nbp));
  if (!lv.P.set || !lv.R.set) return sw_fail(h, "transfer operators of level %d not set", l);
  const bool pre = lv.nu_pre > 0;
  if (pre) {
    SWCHK(zero_vec(h, Xout, lv.n, nbp));
    SWCHK(copy_vec(h, lv.r, Bin, lv.n, nbp));
    SWCHK(mr_smooth(h, lv, Xout, lv.r, lv.nu_pre, nbp));
    SWCHK(launch_ell(h, lv.R, 0, lv.r, nullptr, lc.b, nbp, T_R));
  } else {
    SWCHK(launch_ell(h, lv.R, 0, Bin, nullptr, lc.b, nbp, T_R));
  }
  if (lv.kcycle > 0 && l + 1 < last) {
    SWCHK(ensure_krylov(h, lc.kws, lv.kcycle, lc.n, nbp, false));
    SWCHK(fgmres(h, H, l + 1, lc.b, lc.x, 0.0, lv.kcycle, lv.kcycle, false, lc.kws, nbp, nullptr));
  } else {
    SWCHK(vcycle(h, H, l + 1, lc.b, lc.x, nbp));
  }
  if (pre) SWCHK(launch_ell(h, lv.P, 2, lc.x, Xout, Xout, nbp, T_P));
  else SWCHK(launch_ell(h, lv.P, 0, lc.x, nullptr, Xout, nbp, T_P));
  if (lv.nu_post > 0) {
    SWCHK(apply_op(h, lv, 1, Xout, Bin, lv.r, nbp));
    SWCHK(mr_smooth(h, lv, Xout, lv.r, lv.nu_post, nbp));
  }
  return 0;
}

// Even-site group lists of the prolongators of the even-odd smoothed levels (rebuilt after any
// operator of the hierarchy changed).  Lattice level: rows are parity-sorted, the even sites are the
// first half; block levels: the 16-row tiles named by the tile map of the Schur operator.
static int ensure_even_orders(sw_engine* h, Hier& H) {
  if (H.even_valid) return 0;
  for (int l = 0; l < H.nlevels - 1; ++l) {
    Level& lv = H.lv[l];
    EllOp& P = lv.P;
    SWCHK(dev_free(h, P.order_even));
    P.order_even = nullptr;
    P.ngroups_even = 0;
    SWCHK(free_op(h, lv.Re));
    if (lv.stencil && !lv.w_eo.empty() && lv.R.set && lv.R.cols && lv.R.vals) {
      // R restricted to the columns of the even sites (the first n / 2 rows of the level), compacted:
      // the restriction of a vector whose odd half is zero, read from a half-length array
      const EllOp& R = lv.R;
      const size_t ng = (size_t)R.ngroups, K = (size_t)R.K, G = (size_t)R.G;
      std::vector<int> rc(ng * K);
      std::vector<std::complex<double>> rv(ng * K * G);
      HIPCHK(hipMemcpy(rc.data(), R.cols, rc.size() * sizeof(int), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(rv.data(), R.vals, rv.size() * sizeof(cplx), hipMemcpyDeviceToHost));
      std::vector<std::vector<size_t>> keep(ng);
      size_t Ke = 1;
      for (size_t g = 0; g < ng; ++g) {
        for (size_t k = 0; k < K; ++k) {
          if (rc[g * K + k] >= lv.n / 2) continue;
          bool nz = false;
          for (size_t q = 0; q < G; ++q) nz = nz || rv[(g * K + k) * G + q] != std::complex<double>(0.0, 0.0);
          if (nz) keep[g].push_back(k);
        }
        Ke = std::max(Ke, keep[g].size());
      }
      std::vector<int> ec(ng * Ke, 0);
      std::vector<std::complex<double>> evv(ng * Ke * G, std::complex<double>(0.0, 0.0));
      for (size_t g = 0; g < ng; ++g)
        for (size_t i = 0; i < keep[g].size(); ++i) {
          const size_t k = keep[g][i];
          ec[g * Ke + i] = rc[g * K + k];
          for (size_t q = 0; q < G; ++q) evv[(g * Ke + i) * G + q] = rv[(g * K + k) * G + q];
        }
      EllOp& E = lv.Re;
      E.nrows = R.nrows;
      E.ncols = lv.n / 2;
      E.K = (int)Ke;
      E.G = R.G;
      E.ngroups = R.ngroups;
      SWCHK(upload(h, &E.cols, ec.data(), ec.size()));
      SWCHK(upload(h, &E.vals, (const cplx*)evv.data(), evv.size()));
      E.set = true;
    }
    if (!P.set || !P.cols || lv.w_eo.empty() || P.G < 1) continue;
    std::vector<char> even_row_tile;     // per 16-row tile (block levels)
    if (!lv.stencil) {
      const EllOp& S = lv.eo_op[0];
      if (!S.set || !S.bsr_tmap || S.bsr_RT <= 0 || 16 % P.G) continue;
      std::vector<int> tm(S.bsr_RT);
      HIPCHK(hipMemcpy(tm.data(), S.bsr_tmap, tm.size() * sizeof(int), hipMemcpyDeviceToHost));
      even_row_tile.assign((size_t)lv.n / 16, 0);
      for (int t : tm)
        if (t >= 0 && (size_t)t < even_row_tile.size()) even_row_tile[t] = 1;
    } else if ((lv.n / 2) % P.G) {
      continue;
    }
    std::vector<int> ord(P.ngroups);
    if (P.order && h->ell_order)
      HIPCHK(hipMemcpy(ord.data(), P.order, ord.size() * sizeof(int), hipMemcpyDeviceToHost));
    else
      for (int g = 0; g < P.ngroups; ++g) ord[g] = g;
    std::vector<int> ev;
    for (int g : ord) {
      const long long row = (long long)g * P.G;
      const bool even = lv.stencil ? (row < lv.n / 2) : (even_row_tile[row / 16] != 0);
      if (even) ev.push_back(g);
    }
    if (ev.empty()) continue;
    SWCHK(upload(h, &P.order_even, ev.data(), ev.size()));
    P.ngroups_even = (int)ev.size();
  }
  H.even_valid = true;
  return 0;
}

static inline int* sink_slot(sw_engine* h);
static bool f32_capable(const Hier& H, int level);

// The K-cycle's inner iteration -- L steps of flexible GMRES from a zero guess, preconditioned by the cycle
// of `level` -- in Gram-matrix form (see fgmres_eo_gram): directions without orthogonalisation, one pass for
// all inner products, X = sum_j y_j z_j.  7 vector passes of BLAS-1 instead of 14 for L = 2.
static int kcycle_gram(sw_engine* h, Hier& H, int level, const cplx* B, cplx* X, int L, KrylovWS& ws, int nbp) {
  Level& lv = H.lv[level];
  const size_t vec = (size_t)lv.n * nbp;
  for (int j = 0; j < L; ++j) {
    const cplx* vin = (j == 0) ? B : ws.V + vec * (j - 1);
    SWCHK(vcycle(h, H, level, vin, ws.Z + vec * j, nbp));
    SWCHK(apply_op(h, lv, 0, ws.Z + vec * j, nullptr, ws.V + vec * j, nbp));
  }
  PtrList pu;
  pu.p[0] = B;
  for (int j = 0; j < L; ++j) pu.p[j + 1] = ws.V + vec * j;
  swk::FgTail tg{};
  tg.kind = SW_TAIL_GRAM;
  tg.s = ws.sc;
  tg.j = L;
  tg.h1 = ws.gram;
  tg.tol = 0.0;
  tg.tol_stop = 0.0;
  tg.iter_base = 0;
  tg.first_cycle = 1;
  tg.notconv = sink_slot(h);
  SWCHK(multigram(h, pu, L + 1, lv.n, nbp, ws.gram, &tg));
  SWCHK(zero_vec(h, X, lv.n, nbp));
  PtrList pz;
  for (int q = 0; q < L; ++q) pz.p[q] = ws.Z + vec * q;
  return multiaxpy(h, pz, L, ws.sc.ys, 1.0, X, X, lv.n, nbp, nullptr);
}

static int coarse_correction(sw_engine* h, Hier& H, int l, int nbp) {
  Level& lv = H.lv[l];
  Level& lc = H.lv[l + 1];
  const int last = H.nlevels - 1;
  if (lv.kcycle > 0 && l + 1 < last) {
    SWCHK(ensure_krylov(h, lc.kws, lv.kcycle, lc.n, nbp, false));
    int P = 0, rpb = 0;
    row_blocking(lc.n, nbp, true, &P, &rpb);
    if (h->gram_cycle && lv.kcycle <= SW_GRAM_MAXL && fused_ok(h, P, nbp) &&
        !(h->precond_f32 && f32_capable(H, l + 1)))
      return kcycle_gram(h, H, l + 1, lc.b, lc.x, lv.kcycle, lc.kws, nbp);
    return fgmres(h, H, l + 1, lc.b, lc.x, 0.0, lv.kcycle, lv.kcycle, false, lc.kws, nbp, nullptr);
  }
  return vcycle(h, H, l + 1, lc.b, lc.x, nbp);
}

static swk::StencilArgs eo_stencil_args(sw_engine* h, Level& lv, int nbp);

// One k_schur_step launch on the stencil level (fp64) on the lattice rows a.row0 .. a.row0 + a.nrows - 1
// (a.nrows = 0: all)
template <int MODE>
static int launch_schur_step(sw_engine* h, swk::StencilArgs& a, const cplx* src, const cplx* bp, cplx* dst,
                             int nbp) {
  const int items = (a.nrows > 0 ? a.nrows : a.L) * (a.L / 2);
  if (h->eo_tile > 0 && (MODE == 0 || MODE == 3) && a.nrows == 0 && a.L % 8 == 0) {
    // LDS-staged halo tiles in rotated coordinates, double-buffered by LDS-DMA: one persistent workgroup per CU
    constexpr int TM = (MODE == 0) ? 0 : 3;
    const int tiles_v = a.L / 8, ntiles = (a.L / 4) * tiles_v;
    swk::StencilArgs at = a;
    at.row0 = h->eo_tile_dbg;      // (the tile kernel takes no row window: the field carries the diagnostic bits)
    const int njobs = ntiles * (nbp / 64);
    int grid = std::min(h->num_cus, njobs);
    grid = std::max(8, (grid + 7) & ~7);
    if (h->eo_tile == 8)
      hipLaunchKernelGGL((swk::k_schur_tile<TM, 8>), dim3(grid), dim3(512), 0, h->stream, src, dst, at, tiles_v,
                         ntiles, njobs);
    else
      hipLaunchKernelGGL((swk::k_schur_tile<TM, 4>), dim3(grid), dim3(256), 0, h->stream, src, dst, at, tiles_v,
                         ntiles, njobs);
    KLAUNCH_CHECK();
    return 0;
  }
  const int bpc = (items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  hipLaunchKernelGGL((swk::k_schur_step<cplx, MODE>), dim3(bpc * (nbp / 64)), dim3(SW_BLOCK), 0, h->stream, src,
                     bp, dst, a, bpc);
  KLAUNCH_CHECK();
  return 0;
}

// strip height of the time-skewed order for nu launches that each reach two lattice rows (0: plain order),
// and the probe width the strips are walked with: all nbp probes at once where strips of at least
// 4 (nu - 1) + 2 rows fit the cache at that width, else 64-probe chunk by chunk (the whole schedule once per
// chunk: the same launches per probe, each a quarter as wide at nbp = 256)
static int skew_height(sw_engine* h, Level& lv, int nu, int nbp, int* width) {
  const int L = lv.L;
  *width = nbp;
  if (h->eo_skew == 0 || nu < 2) return 0;
  auto admissible = [&](int H) { return H > 0 && L % H == 0 && L / H >= 2 && H > 4 * (nu - 1) && !(H & 1); };
  for (int w = h->eo_skew_chunk ? 64 : nbp; w >= 64; w = (w > 64 ? 64 : 0)) {
    const double row_bytes = (double)(L / 2) * 2.0 * sizeof(cplx) * w;          // one lattice row of a half vector
    int H = 0;
    if (h->eo_skew > 0) H = h->eo_skew;
    else if (3.0 * row_bytes * L * ((double)nbp / w) > 208.0e6) {
      H = 1;
      while (2 * H <= L / 2 && 3.0 * row_bytes * (2 * H) <= 208.0e6) H *= 2;
    }
    if (admissible(H)) {
      *width = w;
      return H;
    }
    if (H == 0 || h->eo_skew > 0) break;      // nothing to skew (fits the cache), or a fixed height that is not admissible
  }
  return 0;
}

// The nu Schur steps of an even-odd smoothing pass on the stencil level,
//   x_e <- x_e + w_k (bp - S x_e),  k = 0 .. nu-1,
// ping-ponging between `cur` (the iterate on entry) and `nxt`; *result = the buffer the last step wrote.
// On lattices whose three half vectors (both iterates and bp) exceed the 256 MB Infinity Cache every step
// would stream them from HBM (1024^2, 64 probes: 3.2 GB per step at 4.2 TB/s).  There the steps run in a
// TIME-SKEWED order instead (option eo_skew): the lattice rows are cut into strips of H rows; strip by
// strip, all nu steps are applied before the next strip is touched, step k of a strip working on its rows
// shifted down by 2 k (a Schur step reaches two lattice rows), so that everything step k reads from step
// k-1 has been written already -- by the strip itself or by its predecessor:
//   strip 0          step k : rows [2k, H - 2k)            (a shrinking trapezoid: nothing to its left yet)
//   strip s >= 1     step k : rows [sH - 2k, (s+1)H - 2k)  (parallelograms)
//   closing wedge    step k : rows [L - 2k, L + 2k)        (periodic lattice: what the torus still lacks)
// The two ping-pong buffers suffice (rows of step k and of step k-2 that are still needed never overlap).
// A strip's working set is 3 (H + 4) rows: chosen to fit the cache, only the first and the last touch of a
// strip reach HBM.  Same arithmetic per site in another order: results are bit-identical.
static int schur_steps(sw_engine* h, Level& lv, cplx* cur, cplx* nxt, const cplx* bp, int nbp, cplx** result) {
  swk::StencilArgs a = eo_stencil_args(h, lv, nbp);
  const int nu = (int)lv.w_eo.size();
  const int L = lv.L;
  int c0 = 0, cw = nbp;      // probe columns of the current launches
  auto launch = [&](int k, int row0, int nrows, const cplx* src, cplx* dst) -> int {
    a.w = cplx{lv.w_eo[k].real(), lv.w_eo[k].imag()};
    a.row0 = row0;
    a.nrows = nrows;
    const int items = (nrows > 0 ? nrows : L) * (L / 2);
    LaunchScope ls(h, T_SCHUR);
    // algorithmic bytes: three half-vector rows per even site and spin (x_e, b'_e in, x_e out) + links
    if (h->profiling) h->twork[T_SCHUR] += (double)items * (96.0 * cw + 64.0);
    return launch_schur_step<2>(h, a, src + c0, bp + c0, dst + c0, cw);
  };
  const int H = skew_height(h, lv, nu, nbp, &cw);      // 0 = no skewing
  if (H == 0) {
    for (int k = 0; k < nu; ++k) {
      SWCHK(launch(k, 0, 0, cur, nxt));
      std::swap(cur, nxt);
    }
    *result = cur;
    return 0;
  }
  // buf[k & 1] receives step k; step k reads buf[(k + 1) & 1] (step 0: the iterate on entry, in `cur`)
  cplx* buf[2] = {nxt, cur};
  const int ns = L / H;
  for (c0 = 0; c0 < nbp; c0 += cw) {
    for (int sidx = 0; sidx < ns; ++sidx)
      for (int k = 0; k < nu; ++k) {
        const int row0 = (sidx == 0) ? 2 * k : sidx * H - 2 * k;
        const int nrows = (sidx == 0) ? H - 4 * k : H;
        SWCHK(launch(k, row0, nrows, buf[(k + 1) & 1], buf[k & 1]));
      }
    for (int k = 1; k < nu; ++k) SWCHK(launch(k, L - 2 * k, 4 * k, buf[(k + 1) & 1], buf[k & 1]));
  }
  *result = buf[(nu - 1) & 1];
  return 0;
}

// The same smoothing pass in PRODUCT FORM.  The nu steps above give x + q(S)(b' - S x) with the polynomial
// q(z) = (1 - prod_k (1 - w_k z)) / z of degree nu - 1; written as q(z) = beta prod_j (1 - u_j z) (Level::q_w,
// swp::product_form) the pass becomes
//   v = b' - S x              (k_schur_step<1>: x, b' in, v out     -- 3 half-vector passes)
//   v <- v - u_j S v          (k_schur_step<3>, j = 0 .. nu-3       -- 2 passes each: b' is not read)
//   x <- x + beta (v - u S v) (k_schur_step<4>, the last factor     -- 3 passes, in place in x)
// nu launches as before, 2 nu + 2 passes instead of 3 nu (nu = 8: 18 instead of 24).  Same polynomial, other
// arithmetic: equal to the step form to round-off (tests), not bit for bit.  `x` holds the iterate on entry
// and the smoothed iterate on exit; va, vb are two half-vector scratch arrays.  Time-skewed order as above
// (launch k reads what launch k-1 wrote, two rows further out; x is read by launch 0 and written by the
// last one, whose rows lie behind everything launch 0 still has to read).
static int schur_product_steps(sw_engine* h, Level& lv, cplx* x, cplx* va, cplx* vb, const cplx* bp, int nbp) {
  swk::StencilArgs a = eo_stencil_args(h, lv, nbp);
  const int nu = (int)lv.w_eo.size();
  const int L = lv.L;
  cplx* buf[2] = {va, vb};
  int c0 = 0, cw = nbp;      // probe columns of the current launches
  auto launch = [&](int k, int row0, int nrows) -> int {
    a.row0 = row0;
    a.nrows = nrows;
    const int items = (nrows > 0 ? nrows : L) * (L / 2);
    LaunchScope ls(h, T_SCHUR);
    if (k == 0) {
      if (h->profiling) h->twork[T_SCHUR] += (double)items * (96.0 * cw + 64.0);
      return launch_schur_step<1>(h, a, x + c0, bp + c0, buf[0] + c0, cw);
    }
    const std::complex<double> u = lv.q_w[k - 1];
    a.w = cplx{u.real(), u.imag()};
    if (k < nu - 1) {
      if (h->profiling) h->twork[T_SCHUR] += (double)items * (64.0 * cw + 64.0);
      return launch_schur_step<3>(h, a, buf[(k + 1) & 1] + c0, nullptr, buf[k & 1] + c0, cw);
    }
    a.w2 = cplx{lv.q_beta.real(), lv.q_beta.imag()};
    if (h->profiling) h->twork[T_SCHUR] += (double)items * (96.0 * cw + 64.0);
    return launch_schur_step<4>(h, a, buf[(k + 1) & 1] + c0, x + c0, x + c0, cw);
  };
  const int H = skew_height(h, lv, nu, nbp, &cw);
  if (H == 0) {
    for (int k = 0; k < nu; ++k) SWCHK(launch(k, 0, 0));
    return 0;
  }
  const int ns = L / H;
  for (c0 = 0; c0 < nbp; c0 += cw) {
    for (int sidx = 0; sidx < ns; ++sidx)
      for (int k = 0; k < nu; ++k) {
        const int row0 = (sidx == 0) ? 2 * k : sidx * H - 2 * k;
        const int nrows = (sidx == 0) ? H - 4 * k : H;
        SWCHK(launch(k, row0, nrows));
      }
    for (int k = 1; k < nu; ++k) SWCHK(launch(k, L - 2 * k, 4 * k));
  }
  return 0;
}

// Even-odd post-smoothing of the stencil level (k_schur_step): on entry `start` holds the iterate
// after the coarse correction (only its even half is used), on exit Xout the smoothed iterate.
// `start` and `other` are the ping-pong pair chosen by the caller so that the last step lands in Xout.
static int eo_smooth(sw_engine* h, Level& lv, const cplx* Bin, cplx* start, cplx* other, cplx* Xout,
                     int nbp) {
  swk::StencilArgs a;
  a.L = lv.L;
  a.Vh = lv.L * lv.L / 2;
  a.diag = 4.0 + lv.mass;
  a.U1 = lv.U1;
  a.U2 = lv.U2;
  a.nbp = nbp;
  a.nt_store = 0;
  a.tile_w = lv.L;
  // five lattice rows (2 KiB per site and 64-probe chunk, half the sites) must fit an XCD's L2
  if (lv.L > 256) a.tile_w = (lv.L % 256 == 0) ? 256 : ((lv.L % 64 == 0) ? 64 : lv.L);
  if (h->stencil_tile > 0 && lv.L % h->stencil_tile == 0 && h->stencil_tile % 2 == 0) a.tile_w = h->stencil_tile;
  a.w = cplx{0.0, 0.0};
  const int bpc = (a.Vh + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const double di = 1.0 / a.diag;
  cplx* bp = lv.r;   // b'_e lives in the even half of the level's residual buffer
  const dim3 grid(bpc * (nbp / 64));
  {
    LaunchScope ls(h, T_SCHUR);
    if (h->profiling) h->twork[T_SCHUR] += (double)a.Vh * (96.0 * nbp + 64.0);
    hipLaunchKernelGGL((swk::k_eo_hop<0>), grid, dim3(SW_BLOCK), 0, h->stream, Bin, Bin, bp, a, 1.0, di, bpc);
    KLAUNCH_CHECK();
  }
  cplx* cur = nullptr;
  SWCHK(schur_steps(h, lv, start, other, bp, nbp, &cur));
  {
    LaunchScope ls(h, T_SCHUR);
    if (h->profiling) h->twork[T_SCHUR] += (double)a.Vh * (96.0 * nbp + 64.0);
    hipLaunchKernelGGL((swk::k_eo_hop<1>), grid, dim3(SW_BLOCK), 0, h->stream, Bin, (const cplx*)cur, cur, a,
                       di, di, bpc);
    KLAUNCH_CHECK();
  }
  cplx* last_out = cur;
  if (last_out != Xout) return sw_fail(h, "internal: even-odd smoother ended in the wrong buffer");
  return 0;
}

// cycle with the fixed-polynomial smoother (weights set by sw_set_smoother)
static int vcycle_rich(sw_engine* h, Hier& H, int l, const cplx* Bin, cplx* Xout, int nbp) {
  Level& lv = H.lv[l];
  Level& lc = H.lv[l + 1];
  SWCHK(ensure_level_ws(h, lv, nbp));
  SWCHK(ensure_level_ws(h, lc, nbp));
  SWCHK(ensure_even_orders(h, H));
  if (!lv.P.set || !lv.R.set) return sw_fail(h, "transfer operators of level %d not set", l);
  const size_t npost = lv.w_post.size();
  if (!lv.stencil && lv.eo_op[4].set && lv.eo_op[1].set && lv.eo_op[2].set && lv.eo_op[3].set &&
      lv.w_pre.empty() && h->eo_direct) {
    // exact solve of this block level in even-odd reduced form: b'_e = b_e - F b_o ; x_e = S^-1 b'_e
    // (dense, matrix cores) ; x_o = G b_o - Hb x_e -- four launches, nothing below this level is visited
    SWCHK(launch_bsr(h, lv.eo_op[1], 1, Bin, Bin, lv.r, nbp, T_MVM, cplx{0.0, 0.0}));
    SWCHK(launch_bsr(h, lv.eo_op[4], 0, lv.r, nullptr, Xout, nbp, T_COARSEST, cplx{0.0, 0.0}));
    SWCHK(launch_bsr(h, lv.eo_op[2], 0, Bin, nullptr, Xout, nbp, T_MVM, cplx{0.0, 0.0}));
    return launch_bsr(h, lv.eo_op[3], 1, Xout, Xout, Xout, nbp, T_MVM, cplx{0.0, 0.0});
  }
  cplx* xpre = nullptr;
  if (!lv.w_pre.empty()) {
    SWCHK(rich_steps(h, lv, Bin, Xout, lv.t, lv.w_pre, true, nbp, &xpre));
    SWCHK(apply_op(h, lv, 1, xpre, Bin, lv.r, nbp));
    SWCHK(launch_ell(h, lv.R, 0, lv.r, nullptr, lc.b, nbp, T_R));
  } else {
    SWCHK(launch_ell(h, lv.R, 0, Bin, nullptr, lc.b, nbp, T_R));
  }
  SWCHK(coarse_correction(h, H, l, nbp));
  if (!lv.stencil && !lv.w_eo.empty() && !xpre && lv.eo_op[0].set) {
    // even-odd post-smoother of a block level, all four pieces on the MFMA block-row kernel:
    //   b'_e = b_e - F b_o ;  x_e <- x_e + w_k (b'_e - S x_e) ;  x_o = G b_o - Hb x_e
    const bool odd_steps = (lv.w_eo.size() & 1) != 0;
    cplx* cur = odd_steps ? lv.t : Xout;
    cplx* nxt = odd_steps ? Xout : lv.t;
    SWCHK(launch_ell(h, lv.P, 0, lc.x, nullptr, cur, nbp, T_P, cplx{0.0, 0.0}, true));
    SWCHK(launch_bsr(h, lv.eo_op[1], 1, Bin, Bin, lv.r, nbp, T_MVM, cplx{0.0, 0.0}));
    for (size_t k = 0; k < lv.w_eo.size(); ++k) {
      SWCHK(launch_bsr(h, lv.eo_op[0], 3, cur, lv.r, nxt, nbp, T_MVM,
                       cplx{lv.w_eo[k].real(), lv.w_eo[k].imag()}));
      std::swap(cur, nxt);
    }
    SWCHK(launch_bsr(h, lv.eo_op[2], 0, Bin, nullptr, Xout, nbp, T_MVM, cplx{0.0, 0.0}));
    return launch_bsr(h, lv.eo_op[3], 1, Xout, Xout, Xout, nbp, T_MVM, cplx{0.0, 0.0});
  }
  if (lv.stencil && !lv.w_eo.empty() && !xpre) {
    // even-odd post-smoother: prolongate into the buffer from which nu steps end in Xout
    const bool odd_steps = (lv.w_eo.size() & 1) != 0;
    cplx* start = odd_steps ? lv.t : Xout;
    cplx* other = odd_steps ? Xout : lv.t;
    SWCHK(launch_ell(h, lv.P, 0, lc.x, nullptr, start, nbp, T_P, cplx{0.0, 0.0}, true));
    return eo_smooth(h, lv, Bin, start, other, Xout, nbp);
  }
  // place the prolongated iterate so that the ping-pong launches of the post-smoother end in Xout
  const size_t nlaunch = npost;
  cplx* start = (nlaunch % 2 == 0) ? Xout : lv.t;
  cplx* other = (nlaunch % 2 == 0) ? lv.t : Xout;
  if (xpre) SWCHK(launch_ell(h, lv.P, 2, lc.x, xpre, start, nbp, T_P));
  else SWCHK(launch_ell(h, lv.P, 0, lc.x, nullptr, start, nbp, T_P));
  cplx* res = start;
  SWCHK(rich_steps(h, lv, Bin, start, other, lv.w_post, false, nbp, &res));
  if (res != Xout) SWCHK(copy_vec(h, Xout, res, lv.n, nbp));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Single-precision preconditioner (option precond_f32).  The cycle below is vcycle_rich's even-odd /
// fixed-polynomial cycle on complex64 mirrors of the operators: half the bytes on every HBM-bound
// kernel and the f32 matrix cores (twice the fp64 rate on gfx950) for the block operators.  It is only
// ever used INSIDE an fp64 flexible GMRES, whose residuals, orthogonalisation and true-residual
// verification are untouched, so the solve converges to the same fp64 tolerance.
// Supported: fixed-polynomial (Richardson) levels without pre-smoothing and without K-cycles, the
// stencil level smoothed even-odd -- i.e. the tuned configurations; anything else fails loudly.
// ---------------------------------------------------------------------------------------------
template <class CI, class CO>
static int cast_vec(sw_engine* h, const CI* src, CO* dst, size_t count, int cat = T_OTHER) {
  LaunchScope ls(h, cat);
  hipLaunchKernelGGL((swk::k_cast<CI, CO>), dim3((unsigned)((count + SW_BLOCK - 1) / SW_BLOCK)),
                     dim3(SW_BLOCK), 0, h->stream, src, dst, count);
  KLAUNCH_CHECK();
  return 0;
}

static int mirror_op32(sw_engine* h, EllOp& op) {
  if (!op.set) return 0;
  if (op.vals && !op.vals32) {
    const size_t cnt = (size_t)op.ngroups * op.K * op.G;
    SWCHK(dev_realloc(h, &op.vals32, cnt));
    SWCHK(cast_vec(h, (const cplx*)op.vals, op.vals32, cnt));
  }
  if (op.bsr_vals && op.bsr_KS > 0 && !op.bsr_vals32) {
    const int RT = op.bsr_RT > 0 ? op.bsr_RT : op.nrows / 16;
    const size_t cnt = (size_t)RT * op.bsr_KS * 64;
    SWCHK(dev_realloc(h, &op.bsr_vals32, cnt));
    SWCHK(cast_vec(h, (const cplx*)op.bsr_vals, op.bsr_vals32, cnt));
  }
  return 0;
}

static int drop_op32(sw_engine* h, EllOp& op) {
  SWCHK(dev_free(h, op.vals32));
  op.vals32 = nullptr;
  SWCHK(dev_free(h, op.bsr_vals32));
  op.bsr_vals32 = nullptr;
  return 0;
}

// (re)build every complex64 mirror of a hierarchy after its operators changed
static int ensure_f32(sw_engine* h, Hier& H) {
  if (H.f32_valid) return 0;
  for (int l = 0; l < H.nlevels; ++l) {
    Level& lv = H.lv[l];
    EllOp* ops[8] = {&lv.A, &lv.P, &lv.R, &lv.eo_op[0], &lv.eo_op[1], &lv.eo_op[2], &lv.eo_op[3], &lv.eo_op[4]};
    for (EllOp* op : ops) {
      SWCHK(drop_op32(h, *op));
      SWCHK(mirror_op32(h, *op));
    }
    if (lv.stencil && lv.U1) {
      const size_t V = (size_t)lv.L * lv.L;
      SWCHK(dev_realloc(h, &lv.U1f, V));
      SWCHK(dev_realloc(h, &lv.U2f, V));
      SWCHK(cast_vec(h, (const cplx*)lv.U1, lv.U1f, V));
      SWCHK(cast_vec(h, (const cplx*)lv.U2, lv.U2f, V));
    }
  }
  SWCHK(drop_op32(h, H.cinv));
  SWCHK(mirror_op32(h, H.cinv));
  H.f32_valid = true;
  return 0;
}

static int ensure_level_ws32(sw_engine* h, Level& lv, int nbp) {
  if (lv.ws32_nbp == nbp && lv.b32) return 0;
  const size_t cnt = (size_t)lv.n * nbp;
  SWCHK(dev_realloc(h, &lv.b32, cnt));
  SWCHK(dev_realloc(h, &lv.x32, cnt));
  SWCHK(dev_realloc(h, &lv.r32, cnt));
  SWCHK(dev_realloc(h, &lv.t32, cnt));
  SWCHK(dev_free(h, lv.i32)); lv.i32 = nullptr;   // boundary pair: made on demand for this width
  SWCHK(dev_free(h, lv.o32)); lv.o32 = nullptr;
  lv.ws32_nbp = nbp;
  return 0;
}

static int launch_bsr32(sw_engine* h, const EllOp& op, int mode, const cplxf* X, const cplxf* B, cplxf* Y,
                        int nbp, int cat, cplxf w) {
  if (!op.bsr_vals32) return sw_fail(h, "internal: complex64 mirror of a block-row operator missing");
  const int RT = op.bsr_RT > 0 ? op.bsr_RT : op.nrows / 16;
  // tiles of 16 probes per wave: 4 (64 probes) unless the operator is too small to fill the chip
  int NT = 4;
  if ((long long)RT * (nbp / 64) < 4096) NT = 2;
  if (h->f32_tiles == 1 || h->f32_tiles == 2 || h->f32_tiles == 4) NT = h->f32_tiles;
  const int RBn = (RT + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, NCn = nbp / (16 * NT);
  const int bmap = (cat == T_COARSEST) ? 0 : 1;
  const int n1 = h->hier[h->solver_hid].nlevels > 1 ? h->hier[h->solver_hid].lv[1].n : 0;
  const int cls = cat == T_COARSEST ? T_MFMA_DENSE
                                    : (cat == T_MVM ? (op.nrows == n1 || n1 == 0 ? T_MFMA_OP : T_MFMA_OP2)
                                                    : cat);
  LaunchScope ls(h, cls);
  if (h->profiling) h->twork[cls] += 512.0 * (double)RT * (double)op.bsr_KS * (double)nbp;
  if ((h->f32_splitk == 1 && cat == T_COARSEST) ||
      (h->f32_splitk == 2 && (long long)RT * (nbp / 64) < 4096 && op.bsr_KS >= 16)) {
    // few (tile, chunk) pairs: four waves per pair, each a quarter of the k-steps
    const int NTs = (h->f32_tiles == 1) ? 1 : 2;
    const dim3 gsk(RT * (nbp / (16 * NTs)));
#define SKF_LAUNCH(MD, NTT)                                                                       \
  hipLaunchKernelGGL((swk::k_bsr_mfma_f32_sk<MD, NTT>), gsk, dim3(SW_BLOCK), 0, h->stream,         \
                     (const cplxf*)op.bsr_vals32, (const int*)op.bsr_kcol, op.bsr_KS, RT, X, B, Y, \
                     nbp, w, (const int*)op.bsr_tmap)
#define SKF_MODE(NTT)                      \
  do {                                     \
    if (mode == 0) SKF_LAUNCH(0, NTT);     \
    else if (mode == 1) SKF_LAUNCH(1, NTT); \
    else SKF_LAUNCH(3, NTT);               \
  } while (0)
    if (NTs == 1) SKF_MODE(1);
    else SKF_MODE(2);
#undef SKF_MODE
#undef SKF_LAUNCH
    KLAUNCH_CHECK();
    return 0;
  }
  const dim3 grid(RBn * NCn);
  const int stg = (cat == T_COARSEST) ? h->f32_dense_stages : h->f32_stages;
#define BSRF_LAUNCH(MD, NTT, SG)                                                                  \
  hipLaunchKernelGGL((swk::k_bsr_mfma_f32<MD, NTT, SG>), grid, dim3(SW_BLOCK), 0, h->stream,      \
                     (const cplxf*)op.bsr_vals32, (const int*)op.bsr_kcol, op.bsr_KS, RT, X, B, Y, \
                     nbp, w, bmap, (const int*)op.bsr_tmap)
#define BSRF_STG(MD, NTT)                      \
  do {                                         \
    if (stg >= 8) BSRF_LAUNCH(MD, NTT, 8);     \
    else if (stg >= 4) BSRF_LAUNCH(MD, NTT, 4); \
    else BSRF_LAUNCH(MD, NTT, 2);              \
  } while (0)
#define BSRF_MODE(NTT)                         \
  do {                                         \
    if (mode == 0) BSRF_STG(0, NTT);           \
    else if (mode == 1) BSRF_STG(1, NTT);      \
    else BSRF_STG(3, NTT);                     \
  } while (0)
  if (NT == 4) BSRF_MODE(4);
  else if (NT == 2) BSRF_MODE(2);
  else BSRF_MODE(1);
#undef BSRF_MODE
#undef BSRF_STG
#undef BSRF_LAUNCH
  KLAUNCH_CHECK();
  return 0;
}

static int launch_ell32(sw_engine* h, const EllOp& op, int mode, const cplxf* X, const cplxf* B,
                        cplxf* Y, int nbp, int cat, cplxf w = cplxf{0.f, 0.f}, bool even_only = false) {
  if (!op.set) return sw_fail(h, "operator not set");
  if (op.bsr_KS > 0 && (mode == 0 || mode == 1 || mode == 3) && op.bsr_vals32)
    return launch_bsr32(h, op, mode, X, B, Y, nbp, cat, w);
  if (!op.cols || !op.vals32)
    return sw_fail(h, "internal: complex64 mirror of a grouped-ELL operator missing");
  const bool ev = even_only && op.order_even && h->p_even;
  const int ng = ev ? op.ngroups_even : op.ngroups;
  const int* ord = ev ? op.order_even : (h->ell_order ? op.order : nullptr);
  dim3 grid((ng + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, nbp / 64);
  LaunchScope ls(h, cat);
#define ELLF_CASE(GG)                                                                            \
  case GG:                                                                                       \
    if (mode == 0)                                                                               \
      hipLaunchKernelGGL((swk::k_ell<GG, 0, cplxf>), grid, dim3(SW_BLOCK), 0, h->stream,         \
                         (const int*)op.cols, (const cplxf*)op.vals32, op.K, ng, ord, X, B, Y, nbp, w); \
    else if (mode == 1)                                                                          \
      hipLaunchKernelGGL((swk::k_ell<GG, 1, cplxf>), grid, dim3(SW_BLOCK), 0, h->stream,         \
                         (const int*)op.cols, (const cplxf*)op.vals32, op.K, ng, ord, X, B, Y, nbp, w); \
    else if (mode == 2)                                                                          \
      hipLaunchKernelGGL((swk::k_ell<GG, 2, cplxf>), grid, dim3(SW_BLOCK), 0, h->stream,         \
                         (const int*)op.cols, (const cplxf*)op.vals32, op.K, ng, ord, X, B, Y, nbp, w); \
    else                                                                                         \
      hipLaunchKernelGGL((swk::k_ell<GG, 3, cplxf>), grid, dim3(SW_BLOCK), 0, h->stream,         \
                         (const int*)op.cols, (const cplxf*)op.vals32, op.K, ng, ord, X, B, Y, nbp, w); \
    break;
  switch (op.G) {
    ELLF_CASE(1)
    ELLF_CASE(2)
    ELLF_CASE(4)
    ELLF_CASE(8)
    ELLF_CASE(16)
    default:
      return sw_fail(h, "unsupported ELL group size %d", op.G);
  }
#undef ELLF_CASE
  KLAUNCH_CHECK();
  return 0;
}

// even-odd post-smoothing of the stencil level in complex64 (eo_smooth's twin).  C = cplxf2: two probes
// per lane (16-B accesses, rows of nbp / 2 elements), used whenever nbp is a multiple of 128
template <class C>
// reduced: smoothing of the even-odd reduced system itself (vcycle32_even): Bin IS b'_e (no hop before the
// steps) and only the even half of the iterate is wanted (no hop after them); all arrays half-length
static int eo_smooth32_t(sw_engine* h, Level& lv, const cplxf* Bin_, cplxf* start_, cplxf* other_,
                         cplxf* Xout_, int nbp, bool reduced) {
  const int per = (int)(sizeof(C) / sizeof(cplxf));   // probes per lane
  const C* Bin = (const C*)Bin_;
  C* start = (C*)start_;
  C* other = (C*)other_;
  C* Xout = (C*)Xout_;
  swk::StencilArgsT<C> a;
  a.L = lv.L;
  a.Vh = lv.L * lv.L / 2;
  a.diag = (float)(4.0 + lv.mass);
  a.U1 = lv.U1f;
  a.U2 = lv.U2f;
  a.nbp = nbp / per;
  a.nt_store = 0;
  a.tile_w = lv.L;
  // five lattice rows (1 KiB per site and 64-lane chunk, half the sites) must fit an XCD's L2
  if (lv.L > 256) a.tile_w = (lv.L % 256 == 0) ? 256 : ((lv.L % 64 == 0) ? 64 : lv.L);
  if (h->stencil_tile > 0 && lv.L % h->stencil_tile == 0 && h->stencil_tile % 2 == 0) a.tile_w = h->stencil_tile;
  a.w = cplxf{0.f, 0.f};
  const int bpc = (a.Vh + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const dim3 grid(bpc * (a.nbp / 64));
  const float di = (float)(1.0 / (4.0 + lv.mass));
  const C* bp = reduced ? Bin : (const C*)lv.r32;
  if (!reduced) {
    LaunchScope ls(h, T_SCHUR);
    hipLaunchKernelGGL((swk::k_eo_hop<0, C>), grid, dim3(SW_BLOCK), 0, h->stream, Bin, Bin, (C*)lv.r32, a,
                       1.0f, di, bpc);
    KLAUNCH_CHECK();
  }
  C* cur = start;
  C* nxt = other;
  for (size_t k = 0; k < lv.w_eo.size(); ++k) {
    a.w = cplxf{(float)lv.w_eo[k].real(), (float)lv.w_eo[k].imag()};
    LaunchScope ls(h, T_SCHUR);
    hipLaunchKernelGGL((swk::k_schur_step<C>), grid, dim3(SW_BLOCK), 0, h->stream, (const C*)cur,
                       (const C*)bp, nxt, a, bpc);
    KLAUNCH_CHECK();
    std::swap(cur, nxt);
  }
  if (cur != Xout) return sw_fail(h, "internal: even-odd smoother ended in the wrong buffer");
  if (!reduced) {
    LaunchScope ls(h, T_SCHUR);
    hipLaunchKernelGGL((swk::k_eo_hop<1, C>), grid, dim3(SW_BLOCK), 0, h->stream, Bin,
                       (const C*)Xout, Xout, a, di, di, bpc);
    KLAUNCH_CHECK();
  }
  return 0;
}

static int eo_smooth32(sw_engine* h, Level& lv, const cplxf* Bin, cplxf* start, cplxf* other, cplxf* Xout,
                       int nbp, bool reduced = false) {
  if (nbp % 128 == 0 && h->f32_pairs)
    return eo_smooth32_t<swk::cplxf2>(h, lv, Bin, start, other, Xout, nbp, reduced);
  return eo_smooth32_t<cplxf>(h, lv, Bin, start, other, Xout, nbp, reduced);
}

// Y_e = S X_e on complex64 half vectors (operator of the even-odd reduced system, complex64 Krylov basis)
template <class C>
static int schur_apply32_t(sw_engine* h, Level& lv, const cplxf* X_, cplxf* Y_, int nbp) {
  const int per = (int)(sizeof(C) / sizeof(cplxf));
  swk::StencilArgsT<C> a;
  a.L = lv.L;
  a.Vh = lv.L * lv.L / 2;
  a.diag = (float)(4.0 + lv.mass);
  a.U1 = lv.U1f;
  a.U2 = lv.U2f;
  a.nbp = nbp / per;
  a.nt_store = 0;
  a.tile_w = lv.L;
  if (lv.L > 256) a.tile_w = (lv.L % 256 == 0) ? 256 : ((lv.L % 64 == 0) ? 64 : lv.L);
  if (h->stencil_tile > 0 && lv.L % h->stencil_tile == 0 && h->stencil_tile % 2 == 0) a.tile_w = h->stencil_tile;
  a.w = cplxf{0.f, 0.f};
  const int bpc = (a.Vh + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const dim3 grid(bpc * (a.nbp / 64));
  LaunchScope ls(h, T_SCHUR_OP);
  hipLaunchKernelGGL((swk::k_schur_step<C, 0>), grid, dim3(SW_BLOCK), 0, h->stream, (const C*)X_,
                     (const C*)nullptr, (C*)Y_, a, bpc);
  KLAUNCH_CHECK();
  return 0;
}
static int schur_apply32(sw_engine* h, Level& lv, const cplxf* X, cplxf* Y, int nbp) {
  if (nbp % 128 == 0 && h->f32_pairs) return schur_apply32_t<swk::cplxf2>(h, lv, X, Y, nbp);
  return schur_apply32_t<cplxf>(h, lv, X, Y, nbp);
}

static int vcycle32(sw_engine* h, Hier& H, int l, const cplxf* Bin, cplxf* Xout, int nbp) {
  const int last = H.nlevels - 1;
  if (l == last) {
    if (!H.cinv.set) return sw_fail(h, "coarsest inverse not set");
    return launch_ell32(h, H.cinv, 0, Bin, nullptr, Xout, nbp, T_COARSEST);
  }
  Level& lv = H.lv[l];
  Level& lc = H.lv[l + 1];
  if (!lv.rich || lv.gm_m > 0 || !lv.w_pre.empty())
    return sw_fail(h, "precond_f32 supports fixed-polynomial post-smoothing cycles only "
                      "(level %d is configured otherwise)", l);
  SWCHK(ensure_level_ws32(h, lv, nbp));
  SWCHK(ensure_level_ws32(h, lc, nbp));
  SWCHK(ensure_even_orders(h, H));
  if (!lv.P.set || !lv.R.set) return sw_fail(h, "transfer operators of level %d not set", l);
  if (!lv.stencil && lv.eo_op[4].set && lv.eo_op[4].bsr_vals32 && lv.eo_op[1].set && lv.eo_op[2].set &&
      lv.eo_op[3].set && h->eo_direct) {
    const cplxf z0{0.f, 0.f};
    SWCHK(launch_bsr32(h, lv.eo_op[1], 1, Bin, Bin, lv.r32, nbp, T_MVM, z0));
    SWCHK(launch_bsr32(h, lv.eo_op[4], 0, lv.r32, nullptr, Xout, nbp, T_COARSEST, z0));
    SWCHK(launch_bsr32(h, lv.eo_op[2], 0, Bin, nullptr, Xout, nbp, T_MVM, z0));
    return launch_bsr32(h, lv.eo_op[3], 1, Xout, Xout, Xout, nbp, T_MVM, z0);
  }
  SWCHK(launch_ell32(h, lv.R, 0, Bin, nullptr, lc.b32, nbp, T_R));
  if (lv.kcycle > 0 && l + 1 < last) {
    // K-cycle: the few-step inner FGMRES of the coarse system stays fp64 (its Hessenberg solve and
    // orthogonalisation are precision-sensitive and small); its preconditioner is complex64 again
    const size_t cc = (size_t)lc.n * nbp;
    SWCHK(ensure_level_ws(h, lc, nbp));
    SWCHK(cast_vec(h, (const cplxf*)lc.b32, lc.b, cc, T_AXPY));
    SWCHK(ensure_krylov(h, lc.kws, lv.kcycle, lc.n, nbp, false));
    SWCHK(fgmres(h, H, l + 1, lc.b, lc.x, 0.0, lv.kcycle, lv.kcycle, false, lc.kws, nbp, nullptr, true));
    SWCHK(cast_vec(h, (const cplx*)lc.x, lc.x32, cc, T_AXPY));
  } else {
    SWCHK(vcycle32(h, H, l + 1, lc.b32, lc.x32, nbp));
  }
  const cplxf z{0.f, 0.f};
  if (!lv.stencil && !lv.w_eo.empty() && lv.eo_op[0].set) {
    const bool odd_steps = (lv.w_eo.size() & 1) != 0;
    cplxf* cur = odd_steps ? lv.t32 : Xout;
    cplxf* nxt = odd_steps ? Xout : lv.t32;
    SWCHK(launch_ell32(h, lv.P, 0, lc.x32, nullptr, cur, nbp, T_P, z, true));
    SWCHK(launch_bsr32(h, lv.eo_op[1], 1, Bin, Bin, lv.r32, nbp, T_MVM, z));
    for (size_t k = 0; k < lv.w_eo.size(); ++k) {
      SWCHK(launch_bsr32(h, lv.eo_op[0], 3, cur, lv.r32, nxt, nbp, T_MVM,
                         cplxf{(float)lv.w_eo[k].real(), (float)lv.w_eo[k].imag()}));
      std::swap(cur, nxt);
    }
    SWCHK(launch_bsr32(h, lv.eo_op[2], 0, Bin, nullptr, Xout, nbp, T_MVM, z));
    return launch_bsr32(h, lv.eo_op[3], 1, Xout, Xout, Xout, nbp, T_MVM, z);
  }
  if (lv.stencil) {
    if (lv.w_eo.empty())
      return sw_fail(h, "precond_f32 needs the even-odd smoother on the stencil level (sw_set_eo_smoother)");
    const bool odd_steps = (lv.w_eo.size() & 1) != 0;
    cplxf* start = odd_steps ? lv.t32 : Xout;
    cplxf* other = odd_steps ? Xout : lv.t32;
    SWCHK(launch_ell32(h, lv.P, 0, lc.x32, nullptr, start, nbp, T_P, z, true));
    return eo_smooth32(h, lv, Bin, start, other, Xout, nbp);
  }
  // block level with the plain polynomial post-smoother: x <- x + w_k (b - A x)
  const size_t npost = lv.w_post.size();
  cplxf* cur = (npost % 2 == 0) ? Xout : lv.t32;
  cplxf* nxt = (npost % 2 == 0) ? lv.t32 : Xout;
  SWCHK(launch_ell32(h, lv.P, 0, lc.x32, nullptr, cur, nbp, T_P));
  for (size_t k = 0; k < npost; ++k) {
    SWCHK(launch_ell32(h, lv.A, 3, cur, Bin, nxt, nbp, T_MVM,
                       cplxf{(float)lv.w_post[k].real(), (float)lv.w_post[k].imag()}));
    std::swap(cur, nxt);
  }
  if (cur != Xout) return sw_fail(h, "internal: polynomial smoother ended in the wrong buffer");
  return 0;
}

// can the cycle that starts at `level` run on the complex64 path?  (otherwise the fp64 cycle is used)
static bool f32_capable(const Hier& H, int level) {
  for (int l = level; l < H.nlevels - 1; ++l) {
    const Level& lv = H.lv[l];
    if (!lv.rich || lv.gm_m > 0 || !lv.w_pre.empty()) return false;
    if (lv.stencil && lv.w_eo.empty()) return false;
  }
  return level < H.nlevels - 1;
}

// the preconditioner as the fp64 solver sees it: Xout = (fp64) cycle32((complex64) Bin)
static int vcycle_f32_boundary(sw_engine* h, Hier& H, int l, const cplx* Bin, cplx* Xout, int nbp) {
  SWCHK(ensure_f32(h, H));
  Level& lv = H.lv[l];
  SWCHK(ensure_level_ws32(h, lv, nbp));
  const size_t cnt = (size_t)lv.n * nbp;
  if (!lv.i32) {
    SWCHK(dev_realloc(h, &lv.i32, cnt));
    SWCHK(dev_realloc(h, &lv.o32, cnt));
  }
  SWCHK(cast_vec(h, Bin, lv.i32, cnt, T_AXPY));
  SWCHK(vcycle32(h, H, l, lv.i32, lv.o32, nbp));
  return cast_vec(h, (const cplxf*)lv.o32, Xout, cnt, T_AXPY);
}

// ---------------------------------------------------------------------------------------------
// convergence counters and the FGMRES scalar updates that ride on the reductions
// ---------------------------------------------------------------------------------------------
static int reset_slots(sw_engine* h) {
  HIPCHK(hipMemsetAsync(h->d_notconv, 0, SW_NC_SLOTS * sizeof(int), h->stream));
  h->nc_next = 0;
  return 0;
}
// a zeroed counter for the next scalar update of an outer solve
static int take_slot(sw_engine* h, int** slot) {
  if (h->nc_next >= SW_NC_SLOTS - 1) SWCHK(reset_slots(h));
  *slot = h->d_notconv + h->nc_next++;
  return 0;
}
static inline int* sink_slot(sw_engine* h) { return h->d_notconv + SW_NC_SLOTS - 1; }
// number of probes the update that owned `slot` left above tol_stop (drains the stream)
static int read_slot(sw_engine* h, const int* slot, int* count) {
  HIPCHK(hipMemcpyAsync(h->h_notconv, slot, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  SWCHK(stream_sync(h));
  *count = *h->h_notconv;
  return 0;
}
static swk::FgTail tail_begin(const KrylovWS& ws, int first_cycle) {
  swk::FgTail t{};
  t.kind = SW_TAIL_BEGIN;
  t.s = ws.sc;
  t.h1 = ws.nrm;
  t.first_cycle = first_cycle;
  return t;
}
static swk::FgTail tail_hess(const KrylovWS& ws, int j, bool two_pass, bool pyth, double tol, double tol_stop,
                             int iter_base, int* slot) {
  swk::FgTail t{};
  t.kind = SW_TAIL_HESS;
  t.s = ws.sc;
  t.j = j;
  t.h1 = ws.h1;
  t.h2 = two_pass ? ws.h2 : nullptr;
  t.nrm2 = pyth ? ws.h1 + (size_t)(j + 1) * ws.nbp : ws.nrm;
  t.tol = tol;
  t.tol_stop = tol_stop;
  t.iter_base = iter_base;
  t.pyth = pyth ? 1 : 0;
  t.notconv = slot;
  return t;
}
static swk::FgTail tail_verify(const KrylovWS& ws, double tol, double tol_stop, int* slot) {
  swk::FgTail t{};
  t.kind = SW_TAIL_VERIFY;
  t.s = ws.sc;
  t.h1 = ws.nrm;
  t.tol = tol;
  t.tol_stop = tol_stop;
  t.notconv = slot;
  return t;
}

// ---------------------------------------------------------------------------------------------
// batched right-preconditioned flexible GMRES(m) (MG.solve -> fgmres, multigrid.py:347-366);
// one classical Gram-Schmidt pass per step (two with option cgs2), true-residual verification
//  outer == true : restarted, converges every probe to stop_factor * tol (one host read-back per
//                  iteration once the expected count is near)
//  outer == false: exactly `maxiter` (= m, or m + 1 with `aug`) steps from a zero guess, no host
//                  synchronisation
//  aug (inner, unpreconditioned only): LGMRES's augmentation vector (multigrid.py:393-394, SciPy
//                  lgmres with outer_k >= 1): after the m Arnoldi steps one more step whose direction
//                  is `aug` instead of the next Krylov vector, so the cycle minimises the residual over
//                  K_m(A, r) + span{aug}; ws must have room for m + 1 vectors
// ---------------------------------------------------------------------------------------------
static int fgmres(sw_engine* h, Hier& H, int level, const cplx* B, cplx* X, double tol, int maxiter,
                  int m, bool outer, KrylovWS& ws, int nbp, int* iters_total, bool use_precond,
                  const cplx* aug) {
  Level& lv = H.lv[level];
  const int hid_idx = (int)(&H - &h->hier[0]);
  const int check_from = (outer && h->lazy_sync) ? std::max(0, h->sync_hint[hid_idx][level] - 2) : 0;
  const int n = lv.n;
  const size_t vec = (size_t)n * nbp;
  const bool precond = use_precond && (level < H.nlevels - 1);
  const int tb = 256, tg = (nbp + tb - 1) / tb;
  const double tol_stop = outer ? tol * h->stop_factor : tol;
  if (aug && (outer || precond || ws.m < m + 1)) return sw_fail(h, "internal: augmented cycle misused");
  int done = 0;
  bool first = true;
  bool converged = false;
  if (outer) SWCHK(reset_slots(h));
  SWCHK(zero_vec(h, X, n, nbp));
  const cplx* Rcur = B;
  // single-precision preconditioner: on the lattice level its directions are kept complex64 and its
  // input copy is written by the producer of each Krylov vector; elsewhere cast at the boundary
  const bool pf32 = precond && h->precond_f32 && f32_capable(H, level);
  const bool z32 = pf32 && outer && lv.stencil;
  if (z32) {
    SWCHK(ensure_f32(h, H));
    SWCHK(ensure_level_ws32(h, lv, nbp));
    if (!ws.Z32) {
      SWCHK(dev_realloc(h, &ws.Z32, vec * m));
      SWCHK(dev_realloc(h, &ws.v32, vec));
    }
  }
  const bool k32 = z32 && h->f32_krylov;
  if (k32 && !ws.V32) SWCHK(dev_realloc(h, &ws.V32, vec * m));
  while (done < maxiter && !converged) {
    // beta = ||r||
    {
      PtrList pl;
      pl.p[0] = Rcur;
      const swk::FgTail tbeg = tail_begin(ws, first ? 1 : 0);
      SWCHK(multidot(h, pl, 1, Rcur, n, nbp, ws.nrm, nullptr, nullptr, &tbeg));
    }
    // the basis is kept unnormalised (fg_hess_col): vtilde_0 is the residual where it lies,
    // vtilde_{j+1} the orthogonalised A M vtilde_j -- no normalisation passes over the vectors
    auto vt = [&](int k) -> const cplx* { return k == 0 ? Rcur : ws.V + vec * (k - 1); };
    auto vt32 = [&](int k) -> const cplxf* { return k == 0 ? ws.v32 : ws.V32 + vec * (k - 1); };
    if (z32) SWCHK(cast_vec(h, Rcur, ws.v32, vec, T_AXPY));
    const bool two_pass = h->cgs2 || (!outer && h->inner_cgs2);
    int j = 0;
    const int jmax = std::min(m, maxiter - done) + (aug ? 1 : 0);
    for (; j < jmax; ++j) {
      const cplx* vj = vt(j);
      cplx* zj = ws.Z + vec * j;
      cplx* w = ws.V + vec * j;          // becomes vtilde_{j+1}
      cplxf* zj32 = z32 ? ws.Z32 + vec * j : nullptr;
      const bool aug_step = aug && j == jmax - 1;
      // Last step of a cycle: vtilde_{j+1} is never used (the next cycle starts from the true
      // residual), only h_{j+1,j} is.  With one Gram-Schmidt pass that is
      // sqrt(|w|^2 - sum_k |h_{k,j}|^2): |w|^2 rides along in the same multidot pass over w and the
      // orthogonalisation pass (j + 3 vector passes) is not run at all (fg_hess_col, pyth).
      const bool last = h->pyth_last && !two_pass && j == jmax - 1 && m <= 8;   // (short cycles only:
      // a single-pass basis of 30 vectors has lost too much orthogonality for the identity)
      int* slot = sink_slot(h);
      if (outer) SWCHK(take_slot(h, &slot));
      const swk::FgTail th = tail_hess(ws, j, two_pass, last, tol, tol_stop, done, slot);
      if (k32) {
        // complex64 cycle: vtilde_j -> z_j -> w = A z_j -> orthogonalised vtilde_{j+1}, all stored
        // complex64 (the arithmetic of A z and of the inner products is fp64 on widened operands)
        cplxf* w32 = ws.V32 + vec * j;
        SWCHK(vcycle32(h, H, level, vt32(j), zj32, nbp));
        SWCHK((launch_stencil<cplxf, cplxf>(h, lv, 0, zj32, nullptr, w32, nbp, cplx{0.0, 0.0})));
        swk::PtrListT<cplxf> pv32;
        for (int k = 0; k <= j; ++k) pv32.p[k] = vt32(k);
        pv32.p[j + 1] = w32;
        SWCHK(multidot(h, pv32, last ? j + 2 : j + 1, (const cplxf*)w32, n, nbp, ws.h1, ws.sc.svec, ws.c1,
                       last ? &th : nullptr));
        if (last) {
          // (nothing: h_{j+1,j} comes from the dots alone)
        } else if (two_pass) {
          SWCHK(multiaxpy(h, pv32, j + 1, ws.c1, -1.0, (const cplxf*)w32, w32, n, nbp, nullptr));
          SWCHK(multidot(h, pv32, j + 1, (const cplxf*)w32, n, nbp, ws.h2, ws.sc.svec, ws.c1));
          SWCHK(multiaxpy(h, pv32, j + 1, ws.c1, -1.0, (const cplxf*)w32, w32, n, nbp, ws.nrm, nullptr, &th));
        } else {
          SWCHK(multiaxpy(h, pv32, j + 1, ws.c1, -1.0, (const cplxf*)w32, w32, n, nbp, ws.nrm, nullptr, &th));
        }
      } else {
      if (z32) {
        SWCHK(vcycle32(h, H, level, ws.v32, zj32, nbp));
        SWCHK((launch_stencil<cplxf, cplx>(h, lv, 0, zj32, nullptr, w, nbp, cplx{0.0, 0.0})));
      } else {
        if (aug_step) SWCHK(copy_vec(h, zj, aug, n, nbp));
        else if (pf32) SWCHK(vcycle_f32_boundary(h, H, level, vj, zj, nbp));
        else if (precond) SWCHK(vcycle(h, H, level, vj, zj, nbp));
        else SWCHK(copy_vec(h, zj, vj, n, nbp));
        SWCHK(apply_op(h, lv, 0, zj, nullptr, w, nbp));
      }
      PtrList pv;
      for (int k = 0; k <= j; ++k) pv.p[k] = vt(k);
      pv.p[j + 1] = w;
      // pass 1: raw dots d1 = Vt^H w, coefficients c = svec^2 d1 ; w -= Vt c
      SWCHK(multidot(h, pv, last ? j + 2 : j + 1, w, n, nbp, ws.h1, ws.sc.svec, ws.c1, last ? &th : nullptr));
      if (last) {
        // (nothing: h_{j+1,j} comes from the dots alone)
      } else if (two_pass) {
        SWCHK(multiaxpy(h, pv, j + 1, ws.c1, -1.0, w, w, n, nbp, nullptr));
        // pass 2 (re-orthogonalisation): d2 = Vt^H w ; w -= Vt (svec^2 d2) ; ||w||^2
        SWCHK(multidot(h, pv, j + 1, w, n, nbp, ws.h2, ws.sc.svec, ws.c1));
        SWCHK(multiaxpy(h, pv, j + 1, ws.c1, -1.0, w, w, n, nbp, ws.nrm, z32 ? ws.v32 : nullptr, &th));
      } else {
        SWCHK(multiaxpy(h, pv, j + 1, ws.c1, -1.0, w, w, n, nbp, ws.nrm, z32 ? ws.v32 : nullptr, &th));
      }
      }   // fp64 basis
      // read the flag back from the hinted iteration on, at the end of the budget, and in any
      // case every 8th iteration (a stale hint must never cost more than a few iterations)
      // (tol = 0: a fixed number of iterations, e.g. the relaxation sweeps of the setup -- nothing to poll)
      if (outer && tol_stop > 0.0 && (done + j + 1 >= check_from || done + j + 1 >= maxiter ||
                                      ((done + j + 1) & 7) == 0)) {
        int left = 0;
        SWCHK(read_slot(h, slot, &left));
        if (left == 0) {
          converged = true;
          ++j;
          break;
        }
      }
    }
    const int k = j;  // columns built in this cycle
    {
      LaunchScope ls(h, T_OTHER);
      hipLaunchKernelGGL(swk::k_fg_solve, dim3(tg), dim3(tb), 0, h->stream, ws.sc, k);
      KLAUNCH_CHECK();
    }
    if (z32) {
      swk::PtrListT<cplxf> pz;
      for (int q = 0; q < k; ++q) pz.p[q] = ws.Z32 + vec * q;
      SWCHK(multiaxpy(h, pz, k, ws.sc.ys, 1.0, X, X, n, nbp, nullptr));
    } else {
      PtrList pz;
      for (int q = 0; q < k; ++q) pz.p[q] = ws.Z + vec * q;
      SWCHK(multiaxpy(h, pz, k, ws.sc.ys, 1.0, X, X, n, nbp, nullptr));
    }
    done += k;
    first = false;
    if (!outer) break;
    if (converged && h->verify) {
      // (also at done == maxiter: a solve that is flagged converged on its last permitted
      // iteration is still checked, and reported as not converged when the check fails)
      // true residual of every probe; continue when the Arnoldi recurrence was optimistic
      SWCHK(apply_op(h, lv, 1, X, B, ws.rres, nbp));
      PtrList pr;
      pr.p[0] = ws.rres;
      int* slot = nullptr;
      SWCHK(take_slot(h, &slot));
      const swk::FgTail tv = tail_verify(ws, tol, tol_stop, slot);
      SWCHK(multidot(h, pr, 1, ws.rres, n, nbp, ws.nrm, nullptr, nullptr, &tv));
      int left = 0;
      SWCHK(read_slot(h, slot, &left));
      if (left != 0) {
        converged = false;
        Rcur = ws.rres;
        continue;
      }
    } else if (!converged && done < maxiter) {
      SWCHK(apply_op(h, lv, 1, X, B, ws.rres, nbp));
      Rcur = ws.rres;
    }
  }
  if (iters_total) *iters_total = done;
  if (outer) h->sync_hint[hid_idx][level] = converged ? done : 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// host <-> device vector movement in the reference layout
// ---------------------------------------------------------------------------------------------
static int pack_host(sw_engine* h, Level& lv, int nb, const double* Xhost, cplx* dst, int nbp) {
  const size_t bytes = (size_t)nb * lv.n * sizeof(cplx);
  SWCHK(ensure_stage(h, bytes));
  HIPCHK(hipMemcpyAsync(h->stage, Xhost, bytes, hipMemcpyHostToDevice, h->stream));
  LaunchScope ls(h, T_OTHER);
  hipLaunchKernelGGL(swk::k_pack_c, dim3((lv.n + 63) / 64, nbp / 64), dim3(SW_BLOCK), 0, h->stream,
                     (const cplx*)h->stage, nb, lv.n, (const int*)lv.rowmap, dst, nbp);
  KLAUNCH_CHECK();
  return 0;
}
static int unpack_host(sw_engine* h, Level& lv, int nb, const cplx* src, double* Yhost, int nbp) {
  const size_t bytes = (size_t)nb * lv.n * sizeof(cplx);
  SWCHK(ensure_stage(h, bytes));
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_unpack_c, dim3((lv.n + 63) / 64, nbp / 64), dim3(SW_BLOCK), 0,
                       h->stream, src, nbp, lv.n, (const int*)lv.rowmap, (cplx*)h->stage, nb);
    KLAUNCH_CHECK();
  }
  HIPCHK(hipMemcpyAsync(Yhost, h->stage, bytes, hipMemcpyDeviceToHost, h->stream));
  SWCHK(stream_sync(h));
  return 0;
}

static int check_hier(sw_engine* h, int hid, int level, bool need_ready) {
  if (!h) return 1;
  if (hid < 0 || hid >= SW_MAX_HIER) return sw_fail(h, "hierarchy id %d out of range", hid);
  Hier& H = h->hier[hid];
  if (H.nlevels <= 0) return sw_fail(h, "hierarchy %d not defined", hid);
  if (level < 0 || level >= H.nlevels) return sw_fail(h, "level %d out of range", level);
  if (need_ready && !H.ready) return sw_fail(h, "hierarchy %d not finalised (sw_hier_end)", hid);
  return 0;
}

// two scratch vectors per level for the building-block entry points
static int io_vectors(sw_engine* h, Level& lv, int nbp, cplx** a, cplx** b) {
  SWCHK(ensure_level_ws(h, lv, nbp));
  // the cycle never uses lv.b / lv.x of its own start level, so they are free for I/O
  *a = lv.b;
  *b = lv.x;
  return 0;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* sw_version(void) { return "schwinger-hip 0.1 (gfx950)"; }

int sw_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* sw_last_error(sw_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// Return the device memory the process-wide block pool has parked (dev_free) to the driver.
int sw_pool_trim(void) {
  std::vector<ParkedBlock> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    blocks.swap(g_pool);
    g_pool_bytes = 0;
  }
  int rc = 0;
  for (auto& b : blocks) {
    if (hipSetDevice(b.device) != hipSuccess || hipFree(b.p) != hipSuccess) rc = 1;
  }
  return rc;
}

int sw_create(sw_engine** out, int device_id) {
  if (!out) return 1;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return sw_fail(nullptr, "no HIP device available (%s); the engine has no CPU fallback",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n)
    return sw_fail(nullptr, "device id %d out of range (have %d)", device_id, n);
  sw_engine* h = new sw_engine();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) {
    delete h;
    return sw_fail(nullptr, "hipSetDevice(%d) failed", device_id);
  }
  if (hipStreamCreate(&h->stream) != hipSuccess || hipStreamCreate(&h->gen_stream) != hipSuccess) {
    delete h;
    return sw_fail(nullptr, "hipStreamCreate failed");
  }
  void* q = nullptr;
  void* q2 = nullptr;
  const size_t tick_bytes = (size_t)SW_TICKET_CHUNKS * (SW_RED_MAXGROUPS + 1) * sizeof(int);
  if (hipMalloc(&q, SW_NC_SLOTS * sizeof(int)) != hipSuccess || hipMalloc(&q2, tick_bytes) != hipSuccess ||
      hipMemset(q, 0, SW_NC_SLOTS * sizeof(int)) != hipSuccess || hipMemset(q2, 0, tick_bytes) != hipSuccess ||
      hipHostMalloc((void**)&h->h_notconv, sizeof(int)) != hipSuccess) {
    delete h;
    return sw_fail(nullptr, "allocation of the convergence counters failed");
  }
  h->d_notconv = (int*)q;
  h->d_tick = (int*)q2;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
      h->num_cus = cus;
  }
  *out = h;
  return 0;
}

int sw_destroy(sw_engine* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->gen_stream) (void)hipStreamSynchronize(h->gen_stream);
  for (auto& sl : h->slots)
    if (sl.ready) (void)hipEventDestroy(sl.ready);
  if (h->comm && g_rccl_destroy) g_rccl_destroy(h->comm);
  while (!h->allocs.empty()) (void)dev_free(h, h->allocs.back().first);     // (large blocks: parked for the next engine)
  for (auto& r : h->recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (auto& e : h->evpool) (void)hipEventDestroy(e);
  if (h->d_notconv) (void)hipFree(h->d_notconv);
  if (h->d_tick) (void)hipFree(h->d_tick);
  if (h->h_notconv) (void)hipHostFree(h->h_notconv);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->gen_stream) (void)hipStreamDestroy(h->gen_stream);
  delete h;
  return 0;
}

int sw_hier_begin(sw_engine* h, int hid, int nlevels) {
  if (!h) return 1;
  if (hid < 0 || hid >= SW_MAX_HIER) return sw_fail(h, "hierarchy id %d out of range", hid);
  if (nlevels < 1 || nlevels > SW_MAX_LEVELS) return sw_fail(h, "nlevels %d out of range", nlevels);
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  for (int l = 0; l < SW_MAX_LEVELS; ++l) h->sync_hint[hid][l] = 0;
  // release what the previous definition held
  for (int l = 0; l < SW_MAX_LEVELS; ++l) {
    Level& lv = H.lv[l];
    SWCHK(dev_free(h, lv.U1)); SWCHK(dev_free(h, lv.U2));
    SWCHK(free_op(h, lv.A));
    SWCHK(free_op(h, lv.P));
    SWCHK(free_op(h, lv.R));
    SWCHK(free_op(h, lv.Re));
    SWCHK(free_op(h, lv.dinv));
    for (int q = 0; q < 5; ++q) SWCHK(free_op(h, lv.eo_op[q]));
    SWCHK(dev_free(h, lv.rowmap));
    SWCHK(dev_free(h, lv.b)); SWCHK(dev_free(h, lv.x)); SWCHK(dev_free(h, lv.r));
    SWCHK(dev_free(h, lv.t));
    SWCHK(dev_free(h, lv.U1f)); SWCHK(dev_free(h, lv.U2f));
    SWCHK(dev_free(h, lv.b32)); SWCHK(dev_free(h, lv.x32)); SWCHK(dev_free(h, lv.r32));
    SWCHK(dev_free(h, lv.t32)); SWCHK(dev_free(h, lv.i32)); SWCHK(dev_free(h, lv.o32));
    SWCHK(free_krylov(h, lv.kws));
    SWCHK(free_krylov(h, lv.sws));
    SWCHK(free_krylov(h, lv.gws));
    SWCHK(dev_free(h, lv.g1)); SWCHK(dev_free(h, lv.g2));
    SWCHK(dev_free(h, lv.tv)); SWCHK(dev_free(h, lv.tv2));
    lv = Level();
  }
  SWCHK(free_op(h, H.cinv));
  H.nlevels = nlevels;
  H.ready = false;
  H.f32_valid = H.even_valid = false;
  if (hid == 0) {
    // everything that was sized or indexed by the previous definition of the reference
    // hierarchy: deflation vectors, permutations, rhs maps, probe slots and the probe workspace
    h->kd = 0;
    SWCHK(dev_free(h, h->U));
    h->U = nullptr;
    for (int l = 0; l < SW_MAX_LEVELS; ++l) {
      h->lkd[l] = 0;
      SWCHK(dev_free(h, h->lV[l]));
      h->lV[l] = nullptr;
      SWCHK(dev_free(h, h->perm_src[l]));
      h->perm_src[l] = nullptr;
      SWCHK(free_op(h, h->rhsmap[l]));
    }
    HIPCHK(hipStreamSynchronize(h->gen_stream));
    for (auto& sl : h->slots) {
      SWCHK(dev_free(h, sl.p));
      if (sl.ready) (void)hipEventDestroy(sl.ready);
    }
    h->slots.clear();
    h->pb_slot = -1;
    h->pb_probes = nullptr;
    h->pb_level = -1;
    h->pb_nb = h->pb_nbp = 0;
    h->pb_ws_nbp = 0;
  }
  return 0;
}

int sw_set_lattice(sw_engine* h, int hid, int L, double mass, const double* U1, const double* U2) {
  SWCHK(check_hier(h, hid, 0, false));
  if (L < 2 || (L & 1)) return sw_fail(h, "lattice extent L=%d must be even and >= 2", L);
  if (!U1 || !U2) return sw_fail(h, "null link arrays");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[0];
  h->hier[hid].f32_valid = h->hier[hid].even_valid = false;
  lv.stencil = true;
  lv.L = L;
  lv.mass = mass;
  lv.n = 2 * L * L;
  SWCHK(upload(h, (double**)&lv.U1, U1, (size_t)2 * L * L));
  SWCHK(upload(h, (double**)&lv.U2, U2, (size_t)2 * L * L));
  // natural idx(s,x,y) = s*L*L + y*L + x  ->  internal ((par*Vh + ((y*L+x)>>1))*2 + s)
  const int V = L * L, Vh = V / 2;
  lv.h_rowmap.resize(lv.n);
  for (int s = 0; s < 2; ++s)
    for (int y = 0; y < L; ++y)
      for (int x = 0; x < L; ++x) {
        const int par = (x + y) & 1, sh = (y * L + x) >> 1;
        lv.h_rowmap[s * V + y * L + x] = (par * Vh + sh) * 2 + s;
      }
  SWCHK(upload(h, &lv.rowmap, lv.h_rowmap.data(), lv.h_rowmap.size()));
  return 0;
}

int sw_set_csr(sw_engine* h, int hid, int level, int n, const int64_t* indptr,
               const int32_t* indices, const double* data) {
  SWCHK(check_hier(h, hid, level, false));
  if (n <= 0 || !indptr || !indices || !data) return sw_fail(h, "sw_set_csr: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  h->hier[hid].f32_valid = h->hier[hid].even_valid = false;
  if (lv.stencil) {
    if (lv.n != n) return sw_fail(h, "sw_set_csr: n=%d differs from the lattice size %d", n, lv.n);
    return 0;  // the stencil is the operator; the CSR is redundant
  }
  if (lv.n && lv.n != n) return sw_fail(h, "sw_set_csr: level %d already has n=%d", level, lv.n);
  lv.n = n;
  std::vector<int> none;
  SWCHK(free_op(h, lv.A));
  SWCHK(free_op(h, lv.dinv));      // an inverse of the previous operator is stale
  SWCHK(build_ell(h, lv.A, n, n, indptr, indices, (const std::complex<double>*)data, none, none));
  return build_bsr(h, lv.A, n, indptr, indices, (const std::complex<double>*)data, 0.6);
}

int sw_set_transfer(sw_engine* h, int hid, int level, int n_f, int n_c, const int64_t* indptr,
                    const int32_t* indices, const double* data) {
  SWCHK(check_hier(h, hid, level, false));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  if (level + 1 >= H.nlevels) return sw_fail(h, "no level below %d to transfer to", level);
  if (!indptr || !indices || !data) return sw_fail(h, "sw_set_transfer: null arrays");
  HIPCHK(hipSetDevice(h->device));
  Level& lf = H.lv[level];
  Level& lc = H.lv[level + 1];
  if (lf.n == 0) lf.n = n_f;
  if (lc.n == 0) lc.n = n_c;
  if (lf.n != n_f || lc.n != n_c)
    return sw_fail(h, "sw_set_transfer: shape %dx%d does not match levels (%d,%d)", n_f, n_c, lf.n,
                   lc.n);
  if (!lc.h_rowmap.empty()) return sw_fail(h, "coarse level with a row permutation unsupported");
  const std::complex<double>* cd = (const std::complex<double>*)data;
  // P: rows = fine rows in INTERNAL order
  std::vector<int> rows_int;
  if (!lf.h_rowmap.empty()) {
    rows_int.resize(n_f);
    for (int i = 0; i < n_f; ++i) rows_int[lf.h_rowmap[i]] = i;
  }
  std::vector<int> none;
  SWCHK(build_ell(h, lf.P, n_f, n_c, indptr, indices, cd, rows_int, none, 0, true));
  // R = P^H as CSR (rows = coarse), columns mapped to the fine internal order
  const int64_t nnz = indptr[n_f];
  std::vector<int64_t> rp(n_c + 1, 0);
  for (int64_t q = 0; q < nnz; ++q) rp[indices[q] + 1]++;
  for (int c = 0; c < n_c; ++c) rp[c + 1] += rp[c];
  std::vector<int32_t> ri(nnz);
  std::vector<std::complex<double>> rv(nnz);
  std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
  for (int r = 0; r < n_f; ++r)
    for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) {
      const int64_t pos = fill[indices[q]]++;
      ri[pos] = r;
      rv[pos] = std::conj(cd[q]);
    }
  SWCHK(build_ell(h, lf.R, n_c, n_f, rp.data(), ri.data(), rv.data(), none, lf.h_rowmap));
  return 0;
}

int sw_set_coarsest_inv(sw_engine* h, int hid, int n, const double* dense) {
  SWCHK(check_hier(h, hid, 0, false));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  Level& lv = H.lv[H.nlevels - 1];
  if (!dense || n <= 0) return sw_fail(h, "sw_set_coarsest_inv: bad arguments");
  if (lv.n == 0) lv.n = n;
  if (lv.n != n) return sw_fail(h, "coarsest inverse size %d != level size %d", n, lv.n);
  HIPCHK(hipSetDevice(h->device));
  const std::complex<double>* M = (const std::complex<double>*)dense;
  int G = 1;
  for (int g : {16, 8, 4, 2})
    if (n % g == 0) {
      G = g;
      break;
    }
  const int ng = n / G;
  std::vector<int> hc((size_t)ng * n);
  std::vector<std::complex<double>> hv((size_t)ng * n * G);
  for (int gi = 0; gi < ng; ++gi)
    for (int k = 0; k < n; ++k) {
      hc[(size_t)gi * n + k] = k;
      for (int g = 0; g < G; ++g) hv[((size_t)gi * n + k) * G + g] = M[(size_t)(gi * G + g) * n + k];
    }
  SWCHK(free_op(h, H.cinv));
  EllOp& op = H.cinv;
  op.nrows = op.ncols = n;
  op.K = n;
  op.G = G;
  op.ngroups = ng;
  SWCHK(upload(h, &op.cols, hc.data(), hc.size()));
  SWCHK(upload(h, (std::complex<double>**)&op.vals, hv.data(), hv.size()));
  op.set = true;
  if (n % 16 == 0) {
    // dense -> MFMA block-row form: every 4-column group of every 16-row tile
    const int KS = n / 4, RT = n / 16;
    std::vector<int> kcol((size_t)RT * KS);
    std::vector<std::complex<double>> pk((size_t)RT * KS * 64);
    for (int rt = 0; rt < RT; ++rt)
      for (int ks = 0; ks < KS; ++ks) {
        kcol[(size_t)rt * KS + ks] = ks * 4;
        for (int lane = 0; lane < 64; ++lane)
          pk[((size_t)rt * KS + ks) * 64 + lane] =
              M[(size_t)(rt * 16 + (lane & 15)) * n + ks * 4 + (lane >> 4)];
      }
    SWCHK(upload(h, &op.bsr_kcol, kcol.data(), kcol.size()));
    SWCHK(upload(h, (std::complex<double>**)&op.bsr_vals, pk.data(), pk.size()));
    op.bsr_KS = KS;
    op.dense_uniform = true;
  }
  return 0;
}

int sw_set_cycle(sw_engine* h, int hid, int level, int nu_pre, int nu_post, int kcycle) {
  SWCHK(check_hier(h, hid, level, false));
  if (nu_pre < 0 || nu_post < 0 || kcycle < 0 || kcycle > SW_MAXM)
    return sw_fail(h, "sw_set_cycle: bad parameters");
  Level& lv = h->hier[hid].lv[level];
  lv.nu_pre = nu_pre;
  lv.nu_post = nu_post;
  lv.kcycle = kcycle;
  return 0;
}

int sw_set_smoother(sw_engine* h, int hid, int level, int n_pre, const double* w_pre, int n_post,
                    const double* w_post) {
  SWCHK(check_hier(h, hid, level, false));
  if (n_pre < 0 || n_post < 0 || n_pre > 64 || n_post > 64)
    return sw_fail(h, "sw_set_smoother: bad step counts");
  if ((n_pre > 0 && !w_pre) || (n_post > 0 && !w_post)) return sw_fail(h, "null weights");
  Level& lv = h->hier[hid].lv[level];
  lv.w_pre.clear();
  lv.w_post.clear();
  for (int i = 0; i < n_pre; ++i) lv.w_pre.emplace_back(w_pre[2 * i], w_pre[2 * i + 1]);
  for (int i = 0; i < n_post; ++i) lv.w_post.emplace_back(w_post[2 * i], w_post[2 * i + 1]);
  lv.rich = (n_pre + n_post) > 0;
  return 0;
}

// ---- GPU-side setup of a hierarchy ------------------------------------------
int sw_setup_testvectors(sw_engine* h, int hid, int level, int nvec, uint64_t seed, int sweeps,
                         double tol, int maxiter, int precond, int32_t* iters_out) {
  SWCHK(check_hier(h, hid, level, precond != 0));
  if (nvec != SW_TV) return sw_fail(h, "the device setup works with %d test vectors per half", SW_TV);
  if (sweeps < 0 || maxiter < 1) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  Level& lv = H.lv[level];
  if (lv.n <= 0 || (!lv.stencil && !lv.A.set)) return sw_fail(h, "level %d has no operator yet", level);
  const int nbp = 64;
  const size_t cnt = (size_t)lv.n * nbp;
  if (!lv.tv2) SWCHK(dev_realloc(h, &lv.tv2, cnt));
  if (!lv.tv || seed != 0) {
    if (!lv.tv) SWCHK(dev_realloc(h, &lv.tv, cnt));
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_fill_random, dim3((lv.n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, lv.tv, lv.n, nbp, nvec,
                       (unsigned long long)(seed ? seed : 7));
    KLAUNCH_CHECK();
  }
  // the solver's own restart in both modes: ONE Krylov workspace (2 m + 2 vectors of 64 columns, tens
  // of GB on a 1024^2 lattice) serves the setup and the solves that follow
  const int m = std::min(h->restart, maxiter);
  for (int sw = 0; sw < sweeps; ++sw) {
    SWCHK(ensure_krylov(h, lv.sws, m, lv.n, nbp, true));
    int total = 0;
    // V <- A_l^-1 V (inexact): inverse iteration amplifies the near-kernel the coarse space must hold
    SWCHK(fgmres(h, H, level, lv.tv, lv.tv2, tol, maxiter, m, true, lv.sws, nbp, &total, precond != 0));
    SWCHK(stream_sync(h));
    std::swap(lv.tv, lv.tv2);
    if (iters_out) iters_out[sw] = total;
  }
  return 0;
}

int sw_setup_transfer(sw_engine* h, int hid, int level, int nblocks, int rpb, const int32_t* blk_rows,
                      int G, int K, const int32_t* pcols, const int64_t* pmap, const int32_t* porder) {
  SWCHK(check_hier(h, hid, level, false));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  if (level + 1 >= H.nlevels) return sw_fail(h, "no level below %d", level);
  if (!blk_rows || !pcols || !pmap || nblocks <= 0 || rpb <= 0) return sw_fail(h, "bad arguments");
  if (rpb > 512) return sw_fail(h, "aggregate blocks of %d rows exceed the QR kernel's 512", rpb);
  if (G != 1 && G != 2 && G != 4 && G != 8 && G != 16) return sw_fail(h, "bad group size %d", G);
  HIPCHK(hipSetDevice(h->device));
  Level& lf = H.lv[level];
  Level& lc = H.lv[level + 1];
  const int n_f = lf.n, n_c = nblocks * SW_TV;
  if ((long long)nblocks * rpb != n_f) return sw_fail(h, "blocks do not tile level %d", level);
  if (n_f % G) return sw_fail(h, "group size %d does not divide n = %d", G, n_f);
  if (!lf.tv) return sw_fail(h, "level %d has no test vectors (sw_setup_testvectors)", level);
  if (lc.n != 0 && lc.n != n_c) return sw_fail(h, "level %d already has n = %d", level + 1, lc.n);
  if (!lc.h_rowmap.empty()) return sw_fail(h, "coarse level with a row permutation unsupported");
  lc.n = n_c;
  for (size_t i = 0; i < (size_t)nblocks * rpb; ++i)
    if (blk_rows[i] < 0 || blk_rows[i] >= n_f) return sw_fail(h, "block member row out of range");
  const int ng = n_f / G;
  for (size_t i = 0; i < (size_t)ng * K; ++i)
    if (pcols[i] < 0 || pcols[i] >= n_c) return sw_fail(h, "prolongator column out of range");
  const long long qcount = (long long)nblocks * rpb * SW_TV;
  for (size_t i = 0; i < (size_t)ng * K * G; ++i)
    if (pmap[i] >= qcount) return sw_fail(h, "prolongator value map out of range");
  SWCHK(free_op(h, lf.R));
  SWCHK(free_op(h, lf.P));
  // R = P^H in grouped-ELL form: group = block, G = SW_TV rows (the block's coarse dofs), K = rpb
  EllOp& R = lf.R;
  R.nrows = n_c;
  R.ncols = n_f;
  R.K = rpb;
  R.G = SW_TV;
  R.ngroups = nblocks;
  SWCHK(upload(h, &R.cols, (const int*)blk_rows, (size_t)nblocks * rpb));
  SWCHK(dev_realloc(h, &R.vals, (size_t)qcount));
  cplx* Q = nullptr;
  SWCHK(dev_realloc(h, &Q, (size_t)qcount));
  {
    LaunchScope ls(h, T_OTHER);
    dim3 grid((nblocks + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK);
#define QR_LAUNCH(RPL)                                                                           \
  hipLaunchKernelGGL((swk::k_block_qr<RPL>), grid, dim3(SW_BLOCK), 0, h->stream,                 \
                     (const cplx*)lf.tv, 64, (const int*)R.cols, nblocks, rpb, Q, R.vals)
    if (rpb <= 64) QR_LAUNCH(1);
    else if (rpb <= 128) QR_LAUNCH(2);
    else if (rpb <= 256) QR_LAUNCH(4);
    else QR_LAUNCH(8);      // 8 x 8 aggregates of a block level (BASELINE config 5's 3-level hierarchy)
#undef QR_LAUNCH
    KLAUNCH_CHECK();
  }
  R.set = true;
  // P: structure from the host (geometry), values gathered from Q
  EllOp& P = lf.P;
  P.nrows = n_f;
  P.ncols = n_c;
  P.K = K;
  P.G = G;
  P.ngroups = ng;
  SWCHK(upload(h, &P.cols, (const int*)pcols, (size_t)ng * K));
  if (porder) {
    std::vector<char> seen(ng, 0);
    for (int i = 0; i < ng; ++i) {
      if (porder[i] < 0 || porder[i] >= ng || seen[porder[i]]) return sw_fail(h, "porder is not a permutation");
      seen[porder[i]] = 1;
    }
    SWCHK(upload(h, &P.order, (const int*)porder, (size_t)ng));
  }
  const size_t pv = (size_t)ng * K * G;
  SWCHK(dev_realloc(h, &P.vals, pv));
  long long* dmap = nullptr;
  SWCHK(upload(h, &dmap, (const long long*)pmap, pv));
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_fill_from_map, dim3(4096), dim3(SW_BLOCK), 0, h->stream,
                       (const long long*)dmap, (const cplx*)Q, P.vals, pv);
    KLAUNCH_CHECK();
  }
  P.set = true;
  // the coarse image of the test vectors is the starting guess one level down
  if (!lc.tv) SWCHK(dev_realloc(h, &lc.tv, (size_t)n_c * 64));
  SWCHK(launch_ell(h, R, 0, lf.tv, nullptr, lc.tv, 64, T_R));
  SWCHK(stream_sync(h));
  SWCHK(dev_free(h, dmap));
  SWCHK(dev_free(h, Q));
  return 0;
}

int sw_setup_galerkin(sw_engine* h, int hid, int level, int Lc, const int32_t* nbr) {
  SWCHK(check_hier(h, hid, level, false));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  if (level + 1 >= H.nlevels) return sw_fail(h, "no level below %d", level);
  if (!nbr || Lc < 4 || (Lc & 3)) return sw_fail(h, "coarse lattice extent %d must be a multiple of 4", Lc);
  if (!h->use_mfma || !h->mfma_ops) return sw_fail(h, "the device setup needs use_mfma = mfma_ops = 1");
  HIPCHK(hipSetDevice(h->device));
  Level& lf = H.lv[level];
  Level& lc = H.lv[level + 1];
  const int ncs = Lc * Lc, n_c = ncs * 16, n_f = lf.n;
  if (lc.n != n_c) return sw_fail(h, "level %d has n = %d, expected %d sites x 16", level + 1, lc.n, ncs);
  if (!lf.P.set || !lf.R.set) return sw_fail(h, "transfer operators of level %d not set", level);
  for (int i = 0; i < ncs * 5; ++i)
    if (nbr[i] < 0 || nbr[i] >= ncs) return sw_fail(h, "neighbour site out of range");
  for (int I = 0; I < ncs; ++I)
    for (int a = 1; a < 5; ++a)
      for (int b = 0; b < a; ++b)
        if (nbr[I * 5 + a] == nbr[I * 5 + b]) return sw_fail(h, "neighbour lists must hold five distinct sites");
  const int nbp = 256;
  cplx *E = nullptr, *X = nullptr, *Y = nullptr;
  SWCHK(dev_realloc(h, &E, (size_t)n_c * nbp));
  SWCHK(dev_realloc(h, &X, (size_t)n_f * nbp));
  SWCHK(dev_realloc(h, &Y, (size_t)n_f * nbp));
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_probe_unit, dim3((n_c + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, E, Lc, nbp);
    KLAUNCH_CHECK();
  }
  SWCHK(launch_ell(h, lf.P, 0, E, nullptr, X, nbp, T_P));          // X = P E
  SWCHK(apply_op(h, lf, 0, X, nullptr, Y, nbp));                   // Y = A X
  SWCHK(launch_ell(h, lf.R, 0, Y, nullptr, E, nbp, T_R));          // Z = R Y  (over E)
  int* dnbr = nullptr;
  SWCHK(upload(h, &dnbr, (const int*)nbr, (size_t)ncs * 5));
  SWCHK(free_op(h, lc.A));
  EllOp& A = lc.A;
  A.nrows = A.ncols = n_c;
  A.bsr_KS = 20;
  SWCHK(dev_realloc(h, &A.bsr_vals, (size_t)ncs * 20 * 64));
  SWCHK(dev_realloc(h, &A.bsr_kcol, (size_t)ncs * 20));
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_bsr_from_probe, dim3((ncs * 20 + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)E, nbp, (const int*)dnbr, Lc,
                       A.bsr_vals, A.bsr_kcol);
    KLAUNCH_CHECK();
  }
  A.set = true;
  SWCHK(stream_sync(h));
  {
    // the k-step columns are 16 nbr + 4 g: the own-site-last property can be read off nbr
    bool last = Lc > 1;
    for (int I = 0; I < ncs && last; ++I) last = nbr[I * 5 + 4] == I;
    A.bsr_diag_last = last;
  }
  SWCHK(dev_free(h, dnbr));
  SWCHK(dev_free(h, E));
  SWCHK(dev_free(h, X));
  SWCHK(dev_free(h, Y));
  return 0;
}

// In-place inverse of the dense row-major n x n matrix D on the device: Gauss-Jordan with partial
// pivoting (the device counterpart of np.linalg.inv at multigrid.py:342-344); hand-written kernels, no
// library handle involved.  Four small launches per pivot step.
// In-place inverse of the dense row-major [n][n] matrix D: Gauss-Jordan with partial pivoting.  Blocked
// (option gj_block, default 32 columns; n a multiple of 64): the pivot steps of a panel update the panel
// only (n x nb), after which the panel is N = G E -- the panel's composite transformation applied to the unit
// columns -- and every other column c becomes a[:, c] (pivot rows zeroed) + N a[pivot rows, c]: one rank-nb
// update on the fp64 matrix cores (k_bsr_mfma3 in residual mode on the negated panel, the matrix as nbp = n
// "probes").  n = 4096: 0.355 s -> see profiles/r03_ab_sessions.txt (r03ab).
static int gj_invert(sw_engine* h, cplx* D, int n) {
  cplx* colk = nullptr;
  cplx* pvinv = nullptr;
  int* pivs = nullptr;
  SWCHK(dev_realloc(h, &colk, (size_t)n));
  SWCHK(dev_realloc(h, &pvinv, (size_t)1));
  SWCHK(dev_realloc(h, &pivs, (size_t)n + 1));
  int* info = pivs + n;
  HIPCHK(hipMemsetAsync(info, 0, sizeof(int), h->stream));
  const int nb = (h->gj_block >= 8 && h->gj_block % 8 == 0 && n % 64 == 0 && n % h->gj_block == 0 &&
                  n >= 4 * h->gj_block && h->use_mfma && h->mfma_3m) ? h->gj_block : 0;
  const dim3 g1((n + SW_BLOCK - 1) / SW_BLOCK);
  auto pivot_step = [&](int k, int c0, int cn) {
    const dim3 gu((cn + 63) / 64, (n + 16 * SW_WAVES_PER_BLOCK - 1) / (16 * SW_WAVES_PER_BLOCK));
    hipLaunchKernelGGL(swk::k_gj_pivot, dim3(1), dim3(1024), 0, h->stream, (const cplx*)D, n, k, pivs, pvinv,
                       info);
    hipLaunchKernelGGL(swk::k_gj_swap_rows, dim3((cn + SW_BLOCK - 1) / SW_BLOCK), dim3(SW_BLOCK), 0, h->stream, D,
                       n, k, (const int*)pivs, c0, cn);
    hipLaunchKernelGGL(swk::k_gj_column_and_scale, g1, dim3(SW_BLOCK), 0, h->stream, D, n, k,
                       (const cplx*)pvinv, colk, c0, cn);
    hipLaunchKernelGGL(swk::k_gj_update, gu, dim3(SW_BLOCK), 0, h->stream, D, n, k, (const cplx*)colk, c0, cn);
    h->launches += 4;
  };
  if (nb == 0) {
    for (int k = 0; k < n; ++k) {
      pivot_step(k, 0, n);
      // keep the queue shallow: thousands of back-to-back launches without a host sync overran
      // rocprofv3's per-dispatch counter buffers (FETCH_SIZE pass, n = 2048: segmentation fault inside
      // the tool); a drain every 256 pivot steps costs nothing measurable
      if ((k & 255) == 255) HIPCHK(hipStreamSynchronize(h->stream));
    }
  } else {
    cplx* T = nullptr;
    EllOp pn;                       // -N of the current panel in block-row form
    pn.nrows = n;
    pn.ncols = nb;
    pn.bsr_KS = nb / 4;
    pn.set = true;
    SWCHK(dev_realloc(h, &T, (size_t)nb * n));
    SWCHK(dev_realloc(h, &pn.bsr_vals, (size_t)(n / 16) * pn.bsr_KS * 64));
    SWCHK(dev_realloc(h, &pn.bsr_kcol, (size_t)(n / 16) * pn.bsr_KS));
    const size_t items = (size_t)(n / 16) * pn.bsr_KS;
    for (int k0 = 0; k0 < n; k0 += nb) {
      if (nb <= 64) {
        for (int k = k0; k < k0 + nb; ++k) {
          hipLaunchKernelGGL(swk::k_gj_pivot_panel, dim3(1), dim3(1024), 0, h->stream, D, n, k, k0, nb, pivs, info);
          hipLaunchKernelGGL(swk::k_gj_update_panel, dim3((n + 16 * SW_WAVES_PER_BLOCK - 1) / (16 * SW_WAVES_PER_BLOCK)),
                             dim3(SW_BLOCK), 0, h->stream, D, n, k, k0, nb);
        }
        h->launches += 2 * nb;
      } else {
        for (int k = k0; k < k0 + nb; ++k) pivot_step(k, k0, nb);
      }
      hipLaunchKernelGGL(swk::k_gj_block_rows, g1, dim3(SW_BLOCK), 0, h->stream, D, n, k0, nb, (const int*)pivs, T);
      hipLaunchKernelGGL(swk::k_gj_panel_to_bsr, dim3((unsigned)((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK)),
                         dim3(SW_BLOCK), 0, h->stream, (const cplx*)D, n, k0, nb, pn.bsr_vals, pn.bsr_kcol);
      KLAUNCH_CHECK();
      h->launches += 2;
      SWCHK(launch_bsr(h, pn, 1, T, D, D, n, T_OTHER, cplx{0.0, 0.0}));
      if (((k0 / nb) & 7) == 7) HIPCHK(hipStreamSynchronize(h->stream));      // (shallow queue, as above)
    }
    SWCHK(stream_sync(h));
    SWCHK(dev_free(h, T));
    SWCHK(free_op(h, pn));
  }
  hipLaunchKernelGGL(swk::k_gj_unpermute, g1, dim3(SW_BLOCK), 0, h->stream, D, n, (const int*)pivs);
  KLAUNCH_CHECK();
  h->launches += 1;
  int hinfo = 0;
  HIPCHK(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  SWCHK(stream_sync(h));
  SWCHK(dev_free(h, colk));
  SWCHK(dev_free(h, pvinv));
  SWCHK(dev_free(h, pivs));
  if (hinfo != 0) return sw_fail(h, "Gauss-Jordan inverse: the matrix is singular");
  return 0;
}

int sw_setup_invert_coarsest(sw_engine* h, int hid) {
  SWCHK(check_hier(h, hid, 0, false));
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  Level& lv = H.lv[H.nlevels - 1];
  const EllOp& A = lv.A;
  if (H.nlevels < 2 || !A.set) return sw_fail(h, "coarsest level has no operator");
  const int n = lv.n;
  if (n % 16) return sw_fail(h, "coarsest size %d is not a multiple of 16", n);
  if (n > 8192) return sw_fail(h, "coarsest size %d too large for the in-engine dense inverse", n);
  cplx* D = nullptr;
  SWCHK(dev_realloc(h, &D, (size_t)n * n));
  HIPCHK(hipMemsetAsync(D, 0, (size_t)n * n * sizeof(cplx), h->stream));
  if (A.bsr_KS > 0 && !A.bsr_tmap) {
    LaunchScope ls(h, T_OTHER);
    const int items = (n / 16) * A.bsr_KS;
    hipLaunchKernelGGL(swk::k_bsr_to_dense, dim3((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)A.bsr_vals, (const int*)A.bsr_kcol,
                       n / 16, A.bsr_KS, n, D, (const int*)nullptr, (const int*)nullptr);
    KLAUNCH_CHECK();
  } else if (A.cols && A.vals) {
    // a coarsest operator handed over as CSR (sw_set_csr: the reference hierarchy's R A P)
    LaunchScope ls(h, T_OTHER);
    const size_t total = (size_t)A.ngroups * A.K * A.G;
    hipLaunchKernelGGL(swk::k_ell_to_dense, dim3((unsigned)((total + SW_BLOCK - 1) / SW_BLOCK)), dim3(SW_BLOCK), 0,
                       h->stream, (const int*)A.cols, (const cplx*)A.vals, A.K, A.G, A.ngroups, n, D);
    KLAUNCH_CHECK();
  } else {
    (void)dev_free(h, D);
    return sw_fail(h, "coarsest operator in neither grouped-ELL nor full block-row form");
  }
  if (gj_invert(h, D, n) != 0) {
    (void)dev_free(h, D);
    return 1;
  }
  SWCHK(free_op(h, H.cinv));
  EllOp& op = H.cinv;
  op.nrows = op.ncols = n;
  const int KS = n / 4, RT = n / 16;
  SWCHK(dev_realloc(h, &op.bsr_vals, (size_t)RT * KS * 64));
  SWCHK(dev_realloc(h, &op.bsr_kcol, (size_t)RT * KS));
  {
    LaunchScope ls(h, T_OTHER);
    const size_t items = (size_t)RT * KS;
    hipLaunchKernelGGL(swk::k_dense_to_bsr, dim3((unsigned)((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK)),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)D, n, op.bsr_vals, op.bsr_kcol,
                       (const int*)nullptr);
    KLAUNCH_CHECK();
  }
  op.bsr_KS = KS;
  op.set = true;
  op.dense_uniform = true;     // k_dense_to_bsr: the column list depends on the k-step only
  SWCHK(stream_sync(h));
  SWCHK(dev_free(h, D));
  return 0;
}

// The dense coarsest inverse the engine holds (set by the caller or formed on the device) as a row-major
// complex128[n*n] host array: multigrid.py:342-344's coarsest_inv for callers that read the attribute
// (stoch_trace.py:428-435 traces it directly).
int sw_get_coarsest_inv(sw_engine* h, int hid, double* dense) {
  SWCHK(check_hier(h, hid, 0, false));
  if (!dense) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  const EllOp& op = H.cinv;
  if (!op.set) return sw_fail(h, "coarsest inverse missing");
  const int n = op.nrows;
  cplx* D = nullptr;
  SWCHK(dev_realloc(h, &D, (size_t)n * n));
  HIPCHK(hipMemsetAsync(D, 0, (size_t)n * n * sizeof(cplx), h->stream));
  if (op.bsr_KS > 0 && !op.bsr_tmap) {
    LaunchScope ls(h, T_OTHER);
    const size_t items = (size_t)(n / 16) * op.bsr_KS;
    hipLaunchKernelGGL(swk::k_bsr_to_dense, dim3((unsigned)((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK)),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)op.bsr_vals, (const int*)op.bsr_kcol, n / 16,
                       op.bsr_KS, n, D, (const int*)nullptr, (const int*)nullptr);
    KLAUNCH_CHECK();
  } else if (op.cols && op.vals) {
    LaunchScope ls(h, T_OTHER);
    const size_t total = (size_t)op.ngroups * op.K * op.G;
    hipLaunchKernelGGL(swk::k_ell_to_dense, dim3((unsigned)((total + SW_BLOCK - 1) / SW_BLOCK)), dim3(SW_BLOCK), 0,
                       h->stream, (const int*)op.cols, (const cplx*)op.vals, op.K, op.G, op.ngroups, n, D);
    KLAUNCH_CHECK();
  } else {
    (void)dev_free(h, D);
    return sw_fail(h, "coarsest inverse in an unknown form");
  }
  HIPCHK(hipMemcpyAsync(dense, D, (size_t)n * n * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
  SWCHK(stream_sync(h));
  SWCHK(dev_free(h, D));
  return 0;
}

static int schur_apply(sw_engine* h, Level& lv, int mode, const cplx* X, const cplx* Bp, cplx* Y, int nbp);
static int dot_into(sw_engine* h, const cplx* A, const cplx* Bv, int n, int nbp, cplx* out);

// The dense inverse of a block level's even-odd Schur complement, formed ON THE DEVICE from the level's
// Schur operator (eo_op[0], 9-point in 16 x 16 blocks over the even sites): S -> dense (ne*16)^2 ->
// Gauss-Jordan -> block-row form over the even sites' tiles = even-odd operator 4, with which the cycle
// solves the level exactly (vcycle_rich) instead of smoothing it.  Counterpart of np.linalg.inv at
// multigrid.py:342-344 one level up.
int sw_setup_direct_level(sw_engine* h, int hid, int level) {
  SWCHK(check_hier(h, hid, level, false));
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  H.f32_valid = false;
  Level& lv = H.lv[level];
  const EllOp& S = lv.eo_op[0];
  if (lv.stencil || !S.set || !S.bsr_tmap || S.bsr_RT <= 0 || !lv.eo_op[1].set || !lv.eo_op[2].set ||
      !lv.eo_op[3].set)
    return sw_fail(h, "level %d has no even-odd operators (sw_setup_eo_operators / sw_set_eo_operator)", level);
  const int ne = S.bsr_RT, n = ne * 16, ns = lv.n / 16;
  if (n > 8192) return sw_fail(h, "Schur complement of %d rows too large for the in-engine dense inverse", n);
  std::vector<int> E(ne), erank(ns, -1);
  HIPCHK(hipMemcpy(E.data(), S.bsr_tmap, (size_t)ne * sizeof(int), hipMemcpyDeviceToHost));
  for (int r = 0; r < ne; ++r) {
    if (E[r] < 0 || E[r] >= ns || erank[E[r]] >= 0) return sw_fail(h, "level %d: bad tile map of the Schur operator", level);
    erank[E[r]] = r;
  }
  std::vector<int> kc((size_t)ne * S.bsr_KS);
  HIPCHK(hipMemcpy(kc.data(), S.bsr_kcol, kc.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int c : kc)
    if (erank[c >> 4] < 0) return sw_fail(h, "level %d: the Schur operator reaches an odd site", level);
  int *d_rank = nullptr, *d_E = nullptr;
  SWCHK(upload(h, &d_rank, erank.data(), erank.size()));
  SWCHK(upload(h, &d_E, E.data(), E.size()));
  cplx* D = nullptr;
  SWCHK(dev_realloc(h, &D, (size_t)n * n));
  HIPCHK(hipMemsetAsync(D, 0, (size_t)n * n * sizeof(cplx), h->stream));
  {
    LaunchScope ls(h, T_OTHER);
    const int items = ne * S.bsr_KS;
    hipLaunchKernelGGL(swk::k_bsr_to_dense, dim3((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)S.bsr_vals, (const int*)S.bsr_kcol, ne,
                       S.bsr_KS, n, D, (const int*)d_rank, (const int*)nullptr);
    KLAUNCH_CHECK();
  }
  if (gj_invert(h, D, n) != 0) {
    (void)dev_free(h, D);
    (void)dev_free(h, d_rank);
    (void)dev_free(h, d_E);
    return 1;
  }
  EllOp& op = lv.eo_op[4];
  SWCHK(free_op(h, op));
  op.nrows = op.ncols = lv.n;
  op.bsr_RT = ne;
  op.bsr_KS = n / 4;
  SWCHK(upload(h, &op.bsr_tmap, E.data(), E.size()));
  SWCHK(dev_realloc(h, &op.bsr_vals, (size_t)ne * op.bsr_KS * 64));
  SWCHK(dev_realloc(h, &op.bsr_kcol, (size_t)ne * op.bsr_KS));
  {
    LaunchScope ls(h, T_OTHER);
    const size_t items = (size_t)ne * op.bsr_KS;
    hipLaunchKernelGGL(swk::k_dense_to_bsr, dim3((unsigned)((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK)),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)D, n, op.bsr_vals, op.bsr_kcol,
                       (const int*)d_E);
    KLAUNCH_CHECK();
  }
  op.set = true;
  op.dense_uniform = true;     // k_dense_to_bsr: the column list depends on the k-step only
  SWCHK(stream_sync(h));
  SWCHK(dev_free(h, D));
  SWCHK(dev_free(h, d_rank));
  SWCHK(dev_free(h, d_E));
  return 0;
}

// Dense inverse of a small level's operator (n <= 8192, n % 16 == 0; grouped-ELL operator from sw_set_csr, or the
// block-row operator of a device-built level), formed on the device and kept as the level's direct solver:
// solves that START at this level (the MLMC coarse solves A_c^-1 R x of utils.py:306-329 on the reference
// hierarchy's small levels, the fine solves of its coarse difference levels) are then two applications of the
// inverse around one residual -- x = A^-1 b, x += A^-1 (b - A x) -- instead of a multigrid-preconditioned FGMRES
// of hundreds of tiny launches.  The refinement step takes the Gauss-Jordan inverse's residual (eps cond(A),
// ~1e-11) below the solver tolerance.  The level keeps its operator, transfers and smoother for cycles
// that merely pass through it.
int sw_setup_level_inverse(sw_engine* h, int hid, int level) {
  SWCHK(check_hier(h, hid, level, false));
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  Level& lv = H.lv[level];
  const int n = lv.n;
  if (lv.stencil) return sw_fail(h, "level %d is the lattice level", level);
  if (n <= 0 || n % 16 || n > 8192) return sw_fail(h, "level %d: size %d not a multiple of 16 or above 8192", level, n);
  const EllOp& A = lv.A;
  if (!A.set) return sw_fail(h, "level %d has no operator", level);
  cplx* D = nullptr;
  SWCHK(dev_realloc(h, &D, (size_t)n * n));
  HIPCHK(hipMemsetAsync(D, 0, (size_t)n * n * sizeof(cplx), h->stream));
  if (A.cols && A.vals) {
    LaunchScope ls(h, T_OTHER);
    const size_t total = (size_t)A.ngroups * A.K * A.G;
    hipLaunchKernelGGL(swk::k_ell_to_dense, dim3((unsigned)((total + SW_BLOCK - 1) / SW_BLOCK)), dim3(SW_BLOCK), 0,
                       h->stream, (const int*)A.cols, (const cplx*)A.vals, A.K, A.G, A.ngroups, n, D);
    KLAUNCH_CHECK();
  } else if (A.bsr_KS > 0 && !A.bsr_tmap) {
    LaunchScope ls(h, T_OTHER);
    const int items = (n / 16) * A.bsr_KS;
    hipLaunchKernelGGL(swk::k_bsr_to_dense, dim3((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)A.bsr_vals, (const int*)A.bsr_kcol, n / 16,
                       A.bsr_KS, n, D, (const int*)nullptr, (const int*)nullptr);
    KLAUNCH_CHECK();
  } else {
    (void)dev_free(h, D);
    return sw_fail(h, "level %d: operator in neither grouped-ELL nor full block-row form", level);
  }
  if (gj_invert(h, D, n) != 0) {
    (void)dev_free(h, D);
    return 1;
  }
  EllOp& op = lv.dinv;
  SWCHK(free_op(h, op));
  op.nrows = op.ncols = n;
  const int KS = n / 4, RT = n / 16;
  SWCHK(dev_realloc(h, &op.bsr_vals, (size_t)RT * KS * 64));
  SWCHK(dev_realloc(h, &op.bsr_kcol, (size_t)RT * KS));
  {
    LaunchScope ls(h, T_OTHER);
    const size_t items = (size_t)RT * KS;
    hipLaunchKernelGGL(swk::k_dense_to_bsr, dim3((unsigned)((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK)),
                       dim3(SW_BLOCK), 0, h->stream, (const cplx*)D, n, op.bsr_vals, op.bsr_kcol,
                       (const int*)nullptr);
    KLAUNCH_CHECK();
  }
  op.bsr_KS = KS;
  op.set = true;
  op.dense_uniform = true;     // k_dense_to_bsr: the column list depends on the k-step only
  H.f32_valid = false;
  SWCHK(stream_sync(h));
  SWCHK(dev_free(h, D));
  return 0;
}

// Arnoldi relation of a level's operator for the smoother polynomial (hierarchy.smoother_weights fits
// the weights 1/theta_k to the harmonic Ritz values of H): `degree` steps from a pseudo-random start
// vector, classical Gram-Schmidt twice, all on the device -- the host receives the (degree+1) x degree
// Hessenberg matrix only (row-major complex128).  which = 0: the level operator A_l; which = 1: the
// even-odd Schur complement S of the level (stencil level: half vectors through k_schur_step; block
// levels: eo_op[0] on the even sites' rows).
int sw_setup_arnoldi(sw_engine* h, int hid, int level, int which, int degree, uint64_t seed, double* Hout) {
  SWCHK(check_hier(h, hid, level, false));
  if (degree < 1 || degree > SW_MAXM || !Hout || (which != 0 && which != 1))
    return sw_fail(h, "sw_setup_arnoldi: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  Level& lv = H.lv[level];
  if (lv.n <= 0 || (!lv.stencil && !lv.A.set)) return sw_fail(h, "level %d has no operator yet", level);
  const bool schur = which == 1;
  if (schur && !lv.stencil && !(lv.eo_op[0].set && lv.eo_op[0].bsr_tmap))
    return sw_fail(h, "level %d has no even-odd Schur operator", level);
  if (schur && lv.stencil && (lv.n % 2)) return sw_fail(h, "odd lattice level");
  const int nbp = 64;
  const int n = (schur && lv.stencil) ? lv.n / 2 : lv.n;     // rows of the vectors the operator acts on
  const size_t vec = (size_t)n * nbp;
  cplx* V = nullptr;      // degree + 1 basis vectors
  cplx* d = nullptr;      // [2][degree + 1][nbp] dots of the two passes, + [nbp] norm
  SWCHK(dev_realloc(h, &V, vec * (degree + 1)));
  SWCHK(dev_realloc(h, &d, (size_t)(2 * (degree + 1) + 1) * nbp));
  cplx* d1 = d;
  cplx* d2 = d + (size_t)(degree + 1) * nbp;
  cplx* nr = d + (size_t)2 * (degree + 1) * nbp;
  auto normalise = [&](cplx* w) -> int {
    LaunchScope ls(h, T_AXPY);
    hipLaunchKernelGGL(swk::k_colscale, dim3(std::min(4096, (n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK), 1),
                       dim3(SW_BLOCK), 0, h->stream, w, (const cplx*)nr, n, nbp);
    KLAUNCH_CHECK();
    return 0;
  };
  auto apply = [&](const cplx* x, cplx* y) -> int {
    if (!schur) return apply_op(h, lv, 0, x, nullptr, y, nbp);
    if (lv.stencil) return schur_apply(h, lv, 0, x, nullptr, y, nbp);
    SWCHK(zero_vec(h, y, n, nbp));
    return launch_bsr(h, lv.eo_op[0], 0, x, nullptr, y, nbp, T_MVM, cplx{0.0, 0.0});
  };
  // start vector: one pseudo-random column (the other 63 stay zero), on the rows the operator acts on
  {
    cplx* tmp = (schur && !lv.stencil) ? V + vec : V;
    {
      LaunchScope ls(h, T_OTHER);
      hipLaunchKernelGGL(swk::k_fill_random, dim3((n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK), dim3(SW_BLOCK),
                         0, h->stream, tmp, n, nbp, 1, (unsigned long long)(seed ? seed : 2024));
      KLAUNCH_CHECK();
    }
    if (schur && !lv.stencil) {
      SWCHK(zero_vec(h, V, n, nbp));
      LaunchScope ls(h, T_OTHER);
      const int items = lv.eo_op[0].bsr_RT * 16;
      hipLaunchKernelGGL(swk::k_copy_tiles, dim3((items + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                         dim3(SW_BLOCK), 0, h->stream, (const cplx*)tmp, (const int*)lv.eo_op[0].bsr_tmap,
                         lv.eo_op[0].bsr_RT, nbp, V);
      KLAUNCH_CHECK();
    }
    SWCHK(dot_into(h, V, V, n, nbp, nr));
    SWCHK(normalise(V));
  }
  std::vector<std::complex<double>> Hm((size_t)(degree + 1) * degree, std::complex<double>(0.0, 0.0));
  std::vector<std::complex<double>> hd((size_t)(2 * (degree + 1) + 1) * nbp);
  for (int j = 0; j < degree; ++j) {
    cplx* w = V + vec * (j + 1);
    SWCHK(apply(V + vec * j, w));
    PtrList pv;
    for (int k = 0; k <= j; ++k) pv.p[k] = V + vec * k;
    SWCHK(multidot(h, pv, j + 1, (const cplx*)w, n, nbp, d1));
    SWCHK(multiaxpy(h, pv, j + 1, d1, -1.0, (const cplx*)w, w, n, nbp, nullptr));
    SWCHK(multidot(h, pv, j + 1, (const cplx*)w, n, nbp, d2));
    SWCHK(multiaxpy(h, pv, j + 1, d2, -1.0, (const cplx*)w, w, n, nbp, nr));
    SWCHK(stream_sync(h));
    HIPCHK(hipMemcpy(hd.data(), d, hd.size() * sizeof(cplx), hipMemcpyDeviceToHost));
    for (int k = 0; k <= j; ++k)
      Hm[(size_t)k * degree + j] = hd[(size_t)k * nbp] + hd[(size_t)(degree + 1 + k) * nbp];
    const double hn = std::sqrt(std::max(0.0, hd[(size_t)2 * (degree + 1) * nbp].real()));
    Hm[(size_t)(j + 1) * degree + j] = hn;
    if (!(hn > 0.0)) break;      // invariant subspace: the remaining columns stay zero
    SWCHK(normalise(w));
  }
  SWCHK(stream_sync(h));
  std::memcpy(Hout, Hm.data(), Hm.size() * sizeof(std::complex<double>));
  SWCHK(dev_free(h, V));
  SWCHK(dev_free(h, d));
  return 0;
}

int sw_get_level_dense(sw_engine* h, int hid, int level, double* dense) {
  SWCHK(check_hier(h, hid, level, false));
  if (!dense) return sw_fail(h, "null output");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  const EllOp& A = lv.A;
  if (!A.set || A.bsr_KS <= 0) return sw_fail(h, "level %d has no block-row operator", level);
  const int n = lv.n, RT = n / 16, KS = A.bsr_KS;
  std::vector<std::complex<double>> vals((size_t)RT * KS * 64);
  std::vector<int> kcol((size_t)RT * KS);
  SWCHK(stream_sync(h));
  HIPCHK(hipMemcpy(vals.data(), A.bsr_vals, vals.size() * sizeof(cplx), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(kcol.data(), A.bsr_kcol, kcol.size() * sizeof(int), hipMemcpyDeviceToHost));
  std::complex<double>* D = (std::complex<double>*)dense;
  std::fill(D, D + (size_t)n * n, std::complex<double>(0, 0));
  for (int rt = 0; rt < RT; ++rt)
    for (int ks = 0; ks < KS; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = rt * 16 + (lane & 15), col = kcol[(size_t)rt * KS + ks] + (lane >> 4);
        D[(size_t)row * n + col] += vals[((size_t)rt * KS + ks) * 64 + lane];
      }
  return 0;
}

int sw_set_gmres_smoother(sw_engine* h, int hid, int level, int m, int cycles) {
  SWCHK(check_hier(h, hid, level, false));
  if (m < 0 || m > SW_MAXM || (m > 0 && cycles < 1) || cycles > 16)
    return sw_fail(h, "sw_set_gmres_smoother: need 0 <= m <= %d and 1 <= cycles <= 16", SW_MAXM);
  Level& lv = h->hier[hid].lv[level];
  lv.gm_m = m;
  lv.gm_cycles = m > 0 ? cycles : 0;
  return 0;
}

int sw_set_eo_operator(sw_engine* h, int hid, int level, int which, int RT, int KS, const int32_t* tmap,
                       const int32_t* kcol, const double* vals) {
  SWCHK(check_hier(h, hid, level, false));
  if (which < 0 || which > 4) return sw_fail(h, "even-odd operator index %d out of [0,4]", which);
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  h->hier[hid].f32_valid = h->hier[hid].even_valid = false;
  if (lv.stencil || lv.n <= 0 || lv.n % 16) return sw_fail(h, "level %d is not a block level", level);
  if (RT <= 0 || KS <= 0 || (KS & 3) || !tmap || !kcol || !vals) return sw_fail(h, "bad arguments");
  const int tiles = lv.n / 16;
  for (int i = 0; i < RT; ++i)
    if (tmap[i] < 0 || tmap[i] >= tiles) return sw_fail(h, "tile map entry out of range");
  for (size_t i = 0; i < (size_t)RT * KS; ++i)
    if (kcol[i] < 0 || kcol[i] + 4 > lv.n) return sw_fail(h, "k-step column out of range");
  EllOp& op = lv.eo_op[which];
  SWCHK(free_op(h, op));
  if (which < 4) SWCHK(free_op(h, lv.eo_op[4]));     // an inverse of the previous S is stale
  op.nrows = op.ncols = lv.n;
  op.bsr_RT = RT;
  op.bsr_KS = KS;
  SWCHK(upload(h, &op.bsr_tmap, (const int*)tmap, (size_t)RT));
  SWCHK(upload(h, &op.bsr_kcol, (const int*)kcol, (size_t)RT * KS));
  SWCHK(upload(h, (std::complex<double>**)&op.bsr_vals, (const std::complex<double>*)vals,
               (size_t)RT * KS * 64));
  op.set = true;
  check_diag_last(op, kcol, tmap);
  if (which == 4) {
    // a dense inverse: every row tile lists the same columns in the same order?
    bool uni = true;
    for (int r = 1; r < RT && uni; ++r) uni = std::memcmp(kcol, kcol + (size_t)r * KS, sizeof(int32_t) * KS) == 0;
    op.dense_uniform = uni;
  }
  return 0;
}

// The four even-odd operators of a block level built ON THE DEVICE from its operator in block-row form
// (5-point stencil of 16 x 16 blocks, own site last -- what sw_setup_galerkin produces):
//   G = D_oo^-1 (k_block_inverse),  F = A_eo G,  Hb = G A_oe,  S = D_ee - F A_oe (k_block_products);
// the index structure is derived here from the operator's own k-step columns.
int sw_setup_eo_operators(sw_engine* h, int hid, int level, int Lc) {
  SWCHK(check_hier(h, hid, level, false));
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  Level& lv = H.lv[level];
  const int ns = Lc * Lc;
  if (lv.stencil || Lc < 8 || (Lc & 1) || lv.n != ns * 16)
    return sw_fail(h, "level %d is not a block level of %d x %d sites (extent even, >= 8)", level, Lc, Lc);
  const EllOp& A = lv.A;
  if (!A.set || A.bsr_KS != 20 || !A.bsr_vals || !A.bsr_kcol || A.bsr_tmap)
    return sw_fail(h, "level %d has no 5-point block-row operator (20 k-steps per site)", level);
  std::vector<int> kc((size_t)ns * 20);
  HIPCHK(hipMemcpy(kc.data(), A.bsr_kcol, kc.size() * sizeof(int), hipMemcpyDeviceToHost));
  // nbr[s][j]: the site block j of row site s acts on; the own site must be block 4
  std::vector<int> nbr((size_t)ns * 5);
  auto parity = [&](int s) { return ((s % Lc) + (s / Lc)) & 1; };
  for (int s = 0; s < ns; ++s)
    for (int j = 0; j < 5; ++j) {
      const int t = kc[(size_t)s * 20 + 4 * j] / 16;
      for (int g = 0; g < 4; ++g)
        if (kc[(size_t)s * 20 + 4 * j + g] != 16 * t + 4 * g)
          return sw_fail(h, "level %d: k-step columns are not whole site blocks", level);
      if (t < 0 || t >= ns || (j == 4 ? t != s : parity(t) == parity(s)))
        return sw_fail(h, "level %d: not a nearest-neighbour operator with the own site last", level);
      nbr[(size_t)s * 5 + j] = t;
    }
  // The row tiles of the four operators are listed -- i.e. WALKED by the block-row kernel, XCD band by XCD band --
  // in lattice tiles of 16 x 16 sites (SW_EO_WALK_TILE) on lattices of 32 x 32 sites and more: a row of S reads the x tiles of nine
  // sites, and in plain lattice order the neighbours in y are a whole lattice row of x tiles apart (128 sites x
  // 32 KiB = 4 MB at 128 probes: the size of an XCD's L2), so they came from the fabric every time -- 2.73 GB
  // fetched per launch of the 262144-row level's Schur step against 0.84 GB algorithmic
  // (profiles/kernel_pmc_1024.json).  An 8 x 8 tile with its halo is 3.2 MB of x.  (Output tiles go through tmap,
  // columns through kcol: the order of the list is free.)
  std::vector<int> walk(ns);
  for (int s = 0; s < ns; ++s) walk[s] = s;
  if (Lc >= 32 && Lc % 16 == 0) {
    const char* te_ = std::getenv("SW_EO_WALK_TILE");
    // (1024^2, 128 probes, strict: lattice order 508.7 probe-samples/s, tiles of 4 / 8 / 16 sites 512.7 / 515.1 / 521.3;
    // the Schur step of the 262144-row level 395 -> 342 us and 2.73 -> 1.90 GB fetched with 8: r04r / r04s)
    const int T = (te_ && std::atoi(te_) > 0 && Lc % std::atoi(te_) == 0) ? std::atoi(te_) : 16, tpr = Lc / T;
    std::stable_sort(walk.begin(), walk.end(), [&](int a, int b) {
      const int ta = ((a / Lc) / T) * tpr + (a % Lc) / T, tb = ((b / Lc) / T) * tpr + (b % Lc) / T;
      return ta < tb;
    });
  }
  std::vector<int> E, O, rank(ns);
  for (int s : walk) {
    std::vector<int>& v = parity(s) ? O : E;
    rank[s] = (int)v.size();
    v.push_back(s);
  }
  const int ne = (int)E.size(), no = (int)O.size();
  // the nine even targets of S, own site LAST (k_bsr_mfma's register shortcut)
  static const int disp[9][2] = {{2, 0}, {-2, 0}, {0, 2}, {0, -2}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}, {0, 0}};
  auto site_at = [&](int s, int dx, int dy) {
    const int x = ((s % Lc) + dx + Lc) % Lc, y = ((s / Lc) + dy + Lc) % Lc;
    return y * Lc + x;
  };
  auto slot_of = [&](int e, int t) {
    for (int q = 0; q < 9; ++q)
      if (site_at(e, disp[q][0], disp[q][1]) == t) return q;
    return -1;
  };
  struct Built {
    std::vector<int> tmap, kcol;
    int KS;
  };
  auto shape = [&](const std::vector<int>& rows, int nblk, auto&& target) {
    Built b;
    b.KS = 4 * nblk;
    b.tmap = rows;
    b.kcol.resize(rows.size() * (size_t)b.KS);
    for (size_t r = 0; r < rows.size(); ++r)
      for (int q = 0; q < nblk; ++q)
        for (int g = 0; g < 4; ++g) b.kcol[r * b.KS + 4 * q + g] = 16 * target((int)r, q) + 4 * g;
    return b;
  };
  Built bS = shape(E, 9, [&](int r, int q) { return site_at(E[r], disp[q][0], disp[q][1]); });
  Built bF = shape(E, 4, [&](int r, int q) { return nbr[(size_t)E[r] * 5 + q]; });
  Built bG = shape(O, 1, [&](int r, int) { return O[r]; });
  Built bH = shape(O, 4, [&](int r, int q) { return nbr[(size_t)O[r] * 5 + q]; });
  Built* built[4] = {&bS, &bF, &bG, &bH};
  SWCHK(free_op(h, lv.eo_op[4]));                      // an inverse of the previous S is stale
  for (int w = 0; w < 4; ++w) {
    EllOp& op = lv.eo_op[w];
    SWCHK(free_op(h, op));
    op.nrows = op.ncols = lv.n;
    op.bsr_RT = (int)built[w]->tmap.size();
    op.bsr_KS = built[w]->KS;
    SWCHK(upload(h, &op.bsr_tmap, built[w]->tmap.data(), built[w]->tmap.size()));
    SWCHK(upload(h, &op.bsr_kcol, built[w]->kcol.data(), built[w]->kcol.size()));
    SWCHK(dev_realloc(h, &op.bsr_vals, (size_t)op.bsr_RT * op.bsr_KS * 64));
    op.set = true;
    check_diag_last(op, built[w]->kcol.data(), built[w]->tmap.data());
  }
  const cplx* Av = A.bsr_vals;
  auto ablk = [](int s, int j) { return ((long long)s * 20 + 4 * j) * 64; };          // block j of row site s in A
  auto oblk = [](int r, int nblk, int q) { return ((long long)r * nblk + q) * 256; };   // block q of row r
  int* info = nullptr;
  SWCHK(dev_realloc(h, &info, (size_t)1));
  HIPCHK(hipMemsetAsync(info, 0, sizeof(int), h->stream));
  // G
  {
    std::vector<long long> so(no), dof(no);
    for (int r = 0; r < no; ++r) {
      so[r] = ablk(O[r], 4);
      dof[r] = oblk(r, 1, 0);
    }
    long long *dso = nullptr, *ddo = nullptr;
    SWCHK(upload(h, &dso, so.data(), so.size()));
    SWCHK(upload(h, &ddo, dof.data(), dof.size()));
    {
      LaunchScope ls(h, T_OTHER);
      hipLaunchKernelGGL(swk::k_block_inverse, dim3(no), dim3(256), 0, h->stream, Av, (const long long*)dso,
                         lv.eo_op[2].bsr_vals, (const long long*)ddo, info);
      KLAUNCH_CHECK();
    }
    SWCHK(stream_sync(h));
    SWCHK(dev_free(h, dso));
    SWCHK(dev_free(h, ddo));
  }
  auto products = [&](const std::vector<int>& ptr, const std::vector<long long>& ao,
                      const std::vector<long long>& bo, const cplx* Ab, const cplx* Bb,
                      const std::vector<long long>& io, double sign, cplx* out,
                      const std::vector<long long>& oo) -> int {
    int* dptr = nullptr;
    long long *dao = nullptr, *dbo = nullptr, *dio = nullptr, *doo = nullptr;
    SWCHK(upload(h, &dptr, ptr.data(), ptr.size()));
    SWCHK(upload(h, &dao, ao.data(), ao.size()));
    SWCHK(upload(h, &dbo, bo.data(), bo.size()));
    SWCHK(upload(h, &dio, io.data(), io.size()));
    SWCHK(upload(h, &doo, oo.data(), oo.size()));
    {
      LaunchScope ls(h, T_OTHER);
      hipLaunchKernelGGL(swk::k_block_products, dim3((unsigned)oo.size()), dim3(256), 0, h->stream,
                         (const int*)dptr, (const long long*)dao, (const long long*)dbo, Ab, Bb,
                         (const long long*)dio, Av, sign, out, (const long long*)doo);
      KLAUNCH_CHECK();
    }
    SWCHK(stream_sync(h));
    SWCHK(dev_free(h, dptr));
    SWCHK(dev_free(h, dao));
    SWCHK(dev_free(h, dbo));
    SWCHK(dev_free(h, dio));
    SWCHK(dev_free(h, doo));
    return 0;
  };
  const cplx* Gv = lv.eo_op[2].bsr_vals;
  // F(e, j) = A(e, j) G(o_j);  Hb(o, j) = G(o) A(o, j)
  {
    std::vector<int> ptr;
    std::vector<long long> ao, bo, io, oo;
    for (int r = 0; r < ne; ++r)
      for (int j = 0; j < 4; ++j) {
        ptr.push_back((int)ao.size());
        ao.push_back(ablk(E[r], j));
        bo.push_back(oblk(rank[nbr[(size_t)E[r] * 5 + j]], 1, 0));
        io.push_back(-1);
        oo.push_back(oblk(r, 4, j));
      }
    ptr.push_back((int)ao.size());
    SWCHK(products(ptr, ao, bo, Av, Gv, io, 1.0, lv.eo_op[1].bsr_vals, oo));
  }
  {
    std::vector<int> ptr;
    std::vector<long long> ao, bo, io, oo;
    for (int r = 0; r < no; ++r)
      for (int j = 0; j < 4; ++j) {
        ptr.push_back((int)ao.size());
        ao.push_back(oblk(r, 1, 0));
        bo.push_back(ablk(O[r], j));
        io.push_back(-1);
        oo.push_back(oblk(r, 4, j));
      }
    ptr.push_back((int)ao.size());
    SWCHK(products(ptr, ao, bo, Gv, Av, io, 1.0, lv.eo_op[3].bsr_vals, oo));
  }
  // S(e, slot) = [slot == own] D(e) - sum over (j, j') with target(o_j, j') in that slot of F(e, j) A(o_j, j')
  {
    std::vector<int> ptr;
    std::vector<long long> ao, bo, io, oo;
    for (int r = 0; r < ne; ++r) {
      std::vector<std::pair<long long, long long>> lists[9];
      for (int j = 0; j < 4; ++j) {
        const int o = nbr[(size_t)E[r] * 5 + j];
        for (int jp = 0; jp < 4; ++jp) {
          const int q = slot_of(E[r], nbr[(size_t)o * 5 + jp]);
          if (q < 0) return sw_fail(h, "level %d: a two-hop target is outside the nine-point pattern", level);
          lists[q].push_back({oblk(r, 4, j), ablk(o, jp)});
        }
      }
      for (int q = 0; q < 9; ++q) {
        ptr.push_back((int)ao.size());
        for (auto& pr : lists[q]) {
          ao.push_back(pr.first);
          bo.push_back(pr.second);
        }
        io.push_back(q == 8 ? ablk(E[r], 4) : -1);
        oo.push_back(oblk(r, 9, q));
      }
    }
    ptr.push_back((int)ao.size());
    SWCHK(products(ptr, ao, bo, lv.eo_op[1].bsr_vals, Av, io, -1.0, lv.eo_op[0].bsr_vals, oo));
  }
  int hinfo = 0;
  HIPCHK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
  SWCHK(dev_free(h, info));
  if (hinfo != 0) return sw_fail(h, "level %d: a diagonal block of an odd site is singular", level);
  return 0;
}

// Y = op X for one of a block level's even-odd operators (which: 0 S, 1 F, 2 G, 3 Hb) on full-length
// level vectors in the reference layout; rows the operator does not write come back zero
int sw_apply_eo_operator(sw_engine* h, int hid, int level, int which, int nb, const double* X, double* Y) {
  SWCHK(check_hier(h, hid, level, false));
  if (which < 0 || which > 4 || nb <= 0 || !X || !Y) return sw_fail(h, "sw_apply_eo_operator: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  if (lv.stencil || !lv.eo_op[which].set) return sw_fail(h, "level %d has no even-odd operator %d", level, which);
  const int nbp = pad64(nb);
  cplx *a, *b;
  SWCHK(io_vectors(h, lv, nbp, &a, &b));
  SWCHK(pack_host(h, lv, nb, X, a, nbp));
  SWCHK(zero_vec(h, b, lv.n, nbp));
  SWCHK(launch_bsr(h, lv.eo_op[which], 0, a, nullptr, b, nbp, T_MVM, cplx{0.0, 0.0}));
  return unpack_host(h, lv, nb, b, Y, nbp);
}

int sw_get_level_bsr(sw_engine* h, int hid, int level, int* KS, int32_t* kcol, double* vals) {
  SWCHK(check_hier(h, hid, level, false));
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  const EllOp& A = lv.A;
  if (!A.set || A.bsr_KS <= 0) return sw_fail(h, "level %d has no block-row operator", level);
  if (KS) *KS = A.bsr_KS;
  if (!kcol || !vals) return 0;     // size query
  SWCHK(stream_sync(h));
  const size_t items = (size_t)(lv.n / 16) * A.bsr_KS;
  HIPCHK(hipMemcpy(kcol, A.bsr_kcol, items * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(vals, A.bsr_vals, items * 64 * sizeof(cplx), hipMemcpyDeviceToHost));
  return 0;
}

int sw_set_eo_smoother(sw_engine* h, int hid, int level, int n_post, const double* w_post) {
  SWCHK(check_hier(h, hid, level, false));
  if (n_post < 0 || n_post > 64 || (n_post > 0 && !w_post)) return sw_fail(h, "sw_set_eo_smoother: bad arguments");
  Level& lv = h->hier[hid].lv[level];
  if (n_post > 0 && !lv.stencil && !(lv.eo_op[0].set && lv.eo_op[1].set && lv.eo_op[2].set && lv.eo_op[3].set))
    return sw_fail(h, "level %d: the even-odd operators are not set (sw_set_eo_operator)", level);
  lv.w_eo.clear();
  for (int i = 0; i < n_post; ++i) lv.w_eo.emplace_back(w_post[2 * i], w_post[2 * i + 1]);
  if (!swp::product_form(lv.w_eo, lv.q_w, lv.q_beta)) lv.q_w.clear();
  if (n_post > 0) lv.rich = true;     // the cycle with fixed weights (vcycle_rich)
  h->hier[hid].even_valid = false;
  return 0;
}

int sw_hier_end(sw_engine* h, int hid) {
  SWCHK(check_hier(h, hid, 0, false));
  Hier& H = h->hier[hid];
  H.f32_valid = H.even_valid = false;
  for (int l = 0; l < H.nlevels; ++l) {
    Level& lv = H.lv[l];
    if (lv.n <= 0) return sw_fail(h, "level %d has no size", l);
    if (l < H.nlevels - 1) {
      if (!lv.stencil && !lv.A.set) return sw_fail(h, "level %d has no operator", l);
      if (!lv.P.set) return sw_fail(h, "level %d has no prolongator", l);
    }
  }
  if (H.nlevels > 1 && !H.cinv.set) return sw_fail(h, "coarsest inverse missing");
  if (H.nlevels == 1 && !H.lv[0].stencil && !H.lv[0].A.set)
    return sw_fail(h, "single-level hierarchy has no operator");
  // the coarsest level may also carry its operator (optional)
  H.ready = true;
  return 0;
}

int sw_set_solver(sw_engine* h, int restart, int solver_hid) {
  if (!h) return 1;
  if (restart < 1 || restart > SW_MAXM) return sw_fail(h, "restart %d out of [1,%d]", restart, SW_MAXM);
  if (solver_hid < 0 || solver_hid >= SW_MAX_HIER) return sw_fail(h, "bad solver hierarchy id");
  h->restart = restart;
  h->solver_hid = solver_hid;
  return 0;
}

int sw_set_option(sw_engine* h, const char* name, double value) {
  if (!h || !name) return 1;
  if (std::strcmp(name, "use_mfma") == 0) {
    h->use_mfma = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "bsr_map") == 0 || std::strcmp(name, "dense_map") == 0 ||
      std::strcmp(name, "bsr_sub") == 0) {
    const int v = (int)value;
    if (name[4] == 's') {
      if (v < 1) return sw_fail(h, "bsr_sub must be >= 1");
      h->bsr_sub = v;
    } else {
      if (v < 0 || v > 3) return sw_fail(h, "%s must be 0..3", name);
      (name[0] == 'd' ? h->dense_map : h->bsr_map) = v;
    }
    return 0;
  }
  if (std::strcmp(name, "bsr_stages") == 0 || std::strcmp(name, "dense_stages") == 0) {
    if (value != 2.0 && value != 4.0 && value != 8.0 && !(value == 16.0 && name[0] == 'd'))
      return sw_fail(h, "%s must be 2, 4 or 8 (dense_stages: or 16)", name);
    (name[0] == 'd' ? h->dense_stages : h->bsr_stages) = (int)value;
    return 0;
  }
  if (std::strcmp(name, "bsr_nt") == 0) {
    h->bsr_nt = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "bsr_xreg") == 0) {
    h->bsr_xreg = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "p_even") == 0) {
    h->p_even = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "ell_order") == 0) {
    h->ell_order = value != 0.0;
    for (int i = 0; i < SW_MAX_HIER; ++i) h->hier[i].even_valid = false;
    return 0;
  }
  if (std::strcmp(name, "bench_what") == 0) {
    if (value < 0.0 || value > 3.0) return sw_fail(h, "bench_what must be 0..3");
    h->bench_what = (int)value;
    return 0;
  }
  if (std::strcmp(name, "bench_mode") == 0) {
    if (value != 0.0 && value != 1.0 && value != 2.0) return sw_fail(h, "bench_mode must be 0, 1 or 2");
    h->bench_mode = (int)value;
    return 0;
  }
  if (std::strcmp(name, "stencil_nt") == 0) {
    h->stencil_nt = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "stencil_spw") == 0) {
    const int v = (int)value;
    if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8) return sw_fail(h, "stencil_spw must be 0,1,2,4,8");
    h->stencil_spw = v;
    return 0;
  }
  if (std::strcmp(name, "stencil_tile") == 0) {
    h->stencil_tile = (int)value;
    return 0;
  }
  if (std::strcmp(name, "cgs2") == 0) {
    h->cgs2 = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "inner_cgs2") == 0) {
    h->inner_cgs2 = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "eo_direct") == 0) {
    h->eo_direct = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "eo_solve") == 0) {
    h->eo_solve = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "pyth_last") == 0) {
    h->pyth_last = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "precond_f32") == 0) {
    h->precond_f32 = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "f32_tiles") == 0) {
    const int v = (int)value;
    if (v != 0 && v != 1 && v != 2 && v != 4) return sw_fail(h, "f32_tiles must be 0, 1, 2 or 4");
    h->f32_tiles = v;
    return 0;
  }
  if (std::strcmp(name, "f32_splitk") == 0) {
    h->f32_splitk = (int)value;
    return 0;
  }
  if (std::strcmp(name, "dot_blocks") == 0) {
    g_dot_blocks = std::max(64, (int)value);
    g_dot_pmax = g_dot_blocks;
    return 0;
  }
  if (std::strcmp(name, "f32_krylov") == 0) {
    h->f32_krylov = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "f32_pairs") == 0) {
    h->f32_pairs = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "f32_stages") == 0) {
    h->f32_stages = (int)value;
    return 0;
  }
  if (std::strcmp(name, "f32_dense_stages") == 0) {
    h->f32_dense_stages = (int)value;
    return 0;
  }
  if (std::strcmp(name, "stop_factor") == 0) {
    if (!(value > 0.0 && value <= 1.0)) return sw_fail(h, "stop_factor must be in (0, 1]");
    h->stop_factor = value;
    return 0;
  }
  if (std::strcmp(name, "fused_reduce") == 0) {
    h->fused_reduce = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "eo_skew") == 0) {
    if (value < -1.0 || value > 65536.0) return sw_fail(h, "eo_skew must be -1 (automatic), 0 (off) or a strip height");
    h->eo_skew = (int)value;
    return 0;
  }
  if (std::strcmp(name, "direct_small") == 0) {
    h->direct_small = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "gram_cycle") == 0) {
    h->gram_cycle = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "lgmres_aug") == 0) {
    h->lgmres_aug = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "verify") == 0) {
    h->verify = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "lazy_sync") == 0) {
    h->lazy_sync = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "mfma_3m") == 0) {
    h->mfma_3m = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "eo_skew_chunk") == 0) {
    h->eo_skew_chunk = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "gj_block") == 0) {
    const int v = (int)value;
    if (v != 0 && (v < 8 || v > 256 || v % 8)) return sw_fail(h, "gj_block must be 0 or a multiple of 8 in 8..256");
    h->gj_block = v;
    return 0;
  }
  if (std::strcmp(name, "dense_lds") == 0) {
    if (value != 0.0 && value != 1.0 && value != 2.0 && value != 4.0) return sw_fail(h, "dense_lds must be 0, 2 or 4");
    h->dense_lds = value == 1.0 ? 2 : (int)value;
    return 0;
  }
  if (std::strcmp(name, "eo_tile_dbg") == 0) {
    h->eo_tile_dbg = (int)value & 3;
    return 0;
  }
  if (std::strcmp(name, "eo_tile") == 0) {
    if (value != 0.0 && value != 4.0 && value != 8.0) return sw_fail(h, "eo_tile must be 0 (off), 4 or 8 waves");
    h->eo_tile = (int)value;
    return 0;
  }
  if (std::strcmp(name, "eo_product") == 0) {
    h->eo_product = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "mfma3_tiles") == 0) {
    const int v = (int)value;
    if (v != 0 && v != 1 && v != 2 && v != 4) return sw_fail(h, "mfma3_tiles must be 0, 1, 2 or 4");
    h->mfma3_tiles = v;
    return 0;
  }
  if (std::strcmp(name, "mfma_ops") == 0) {
    h->mfma_ops = value != 0.0;
    return 0;
  }
  if (std::strcmp(name, "mfma_small_tiles") == 0) {
    if (value != 0.0 && value != 2.0 && value != 4.0)
      return sw_fail(h, "mfma_small_tiles must be 0, 2 or 4");
    h->mfma_small_tiles = (int)value;
    return 0;
  }
  if (std::strcmp(name, "mfma_tiles") == 0) {
    if (value != 2.0 && value != 4.0) return sw_fail(h, "mfma_tiles must be 2 or 4");
    h->mfma_tiles = (int)value;
    return 0;
  }
  return sw_fail(h, "unknown option %s", name);
}

// ---- device eigensolver: block subspace iteration with the engine's own batched solves as the
// shift-invert (the counterpart of eigs(A_l, k, sigma=0) at multigrid.py:174 and of
// eigsh(gamma_3 A, k, sigma=0) at utils.py:140).  The engine supplies the O(n) work on blocks of 64
// vectors -- W = Op^-1 V, the 64 x 64 Gram matrices V^H W, W^H W, block rotations V <- W Y -- the caller
// the 64 x 64 dense algebra in between (Rayleigh-Ritz, Cholesky-QR).
static int solve_dev(sw_engine* h, int hid, int level0, const cplx* B, cplx* X, double tol, int maxiter,
                     int nbp, int* total);
static int eig_check(sw_engine* h, int a) {
  if (h->eig_n <= 0 || !h->eig_buf[0]) return sw_fail(h, "sw_eig_begin has not been called");
  if (a < 0 || a > 2) return sw_fail(h, "block buffer index %d out of 0..2", a);
  return 0;
}

int sw_eig_begin(sw_engine* h, int hid, int level, uint64_t seed) {
  SWCHK(check_hier(h, hid, level, true));
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  if (lv.n <= 0 || (lv.n & 3)) return sw_fail(h, "level %d has n = %d (a multiple of 4 is needed)", level, lv.n);
  const size_t cnt = (size_t)lv.n * 64;
  for (int q = 0; q < 3; ++q) SWCHK(dev_realloc(h, &h->eig_buf[q], cnt));
  SWCHK(dev_realloc(h, &h->eig_small, (size_t)4096));
  // gamma_3 = +1 on the first half of the REFERENCE order, -1 on the second (multigrid.py:130-133)
  std::vector<signed char> sg(lv.n);
  for (int i = 0; i < lv.n; ++i) {
    const int r = lv.h_rowmap.empty() ? i : lv.h_rowmap[i];
    sg[r] = (i < lv.n / 2) ? 1 : -1;
  }
  SWCHK(upload(h, &h->eig_sign, (const signed char*)sg.data(), (size_t)lv.n));
  h->eig_hid = hid;
  h->eig_level = level;
  h->eig_n = lv.n;
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_fill_random, dim3((lv.n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK),
                       dim3(SW_BLOCK), 0, h->stream, h->eig_buf[0], lv.n, 64, 64,
                       (unsigned long long)(seed ? seed : 11));
    KLAUNCH_CHECK();
  }
  return stream_sync(h);
}

int sw_eig_end(sw_engine* h) {
  if (!h) return 1;
  for (int q = 0; q < 3; ++q) {
    if (h->eig_buf[q]) SWCHK(dev_free(h, h->eig_buf[q]));
    h->eig_buf[q] = nullptr;
  }
  if (h->eig_sign) SWCHK(dev_free(h, h->eig_sign));
  h->eig_sign = nullptr;
  if (h->eig_small) SWCHK(dev_free(h, h->eig_small));
  h->eig_small = nullptr;
  h->eig_n = 0;
  h->eig_hid = h->eig_level = -1;
  return 0;
}

// Overwrite the first ncols columns of block buffer `dst` with host vectors (reference order, ncols
// contiguous vectors of length n), e.g. converged vectors of the level above as a start.
int sw_eig_load(sw_engine* h, int dst, int ncols, const double* X) {
  SWCHK(eig_check(h, dst));
  if (ncols < 1 || ncols > 64 || !X) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[h->eig_hid].lv[h->eig_level];
  const int spare = dst == 2 ? 1 : 2;
  // pack into a spare buffer (all 64 columns written, zero beyond ncols), then copy the columns over
  SWCHK(pack_host(h, lv, ncols, X, h->eig_buf[spare], 64));
  HIPCHK(hipMemcpy2DAsync(h->eig_buf[dst], 64 * sizeof(cplx), h->eig_buf[spare], 64 * sizeof(cplx),
                          (size_t)ncols * sizeof(cplx), (size_t)lv.n, hipMemcpyDeviceToDevice, h->stream));
  return stream_sync(h);
}

// dst = Op^-1 src on all 64 columns: mode 0 Op = A_level (multigrid.py:174), mode 1 Op = gamma_3 A_level
// (utils.py:137-140; Q^-1 = A^-1 gamma_3).  Batched solve of the (hierarchy, level) to `tol`.
int sw_eig_solve(sw_engine* h, int src, int dst, int mode, double tol, int maxiter, int32_t* iters_max) {
  SWCHK(eig_check(h, src));
  SWCHK(eig_check(h, dst));
  if (src == dst) return sw_fail(h, "source and destination buffers must differ");
  if (mode != 0 && mode != 1) return sw_fail(h, "mode must be 0 (A) or 1 (gamma_3 A)");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[h->eig_hid];
  Level& lv = H.lv[h->eig_level];
  const cplx* rhs = h->eig_buf[src];
  if (mode == 1) {
    cplx* tmp = h->eig_buf[3 - src - dst];
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_row_sign, dim3((lv.n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, 1),
                       dim3(SW_BLOCK), 0, h->stream, rhs, (const signed char*)h->eig_sign, tmp, lv.n, 64);
    KLAUNCH_CHECK();
    rhs = tmp;
  }
  int total = 0;
  SWCHK(solve_dev(h, h->eig_hid, h->eig_level, rhs, h->eig_buf[dst], tol, maxiter, 64, &total));
  SWCHK(stream_sync(h));
  if (iters_max) *iters_max = total;
  return 0;
}

// out[64*64] (row-major complex128) = buf_a^H buf_b
int sw_eig_gram(sw_engine* h, int a, int b, double* out) {
  SWCHK(eig_check(h, a));
  SWCHK(eig_check(h, b));
  if (!out) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  const int n = h->eig_n;
  int rpb = std::max(64, ((n + 255) / 256 + 3) & ~3);     // ~256 row blocks, multiples of 4 rows
  const int P = (n + rpb - 1) / rpb;
  SWCHK(ensure_partial(h, (size_t)P * 4096 * sizeof(cplx)));
  {
    LaunchScope ls(h, T_DOTS);
    hipLaunchKernelGGL(swk::k_block_gram, dim3(P, 4), dim3(SW_BLOCK), 0, h->stream,
                       (const cplx*)h->eig_buf[a], (const cplx*)h->eig_buf[b], n, rpb, h->partial);
    KLAUNCH_CHECK();
  }
  {
    LaunchScope ls(h, T_DOTS);
    hipLaunchKernelGGL(swk::k_block_gram_reduce, dim3(64), dim3(64), 0, h->stream,
                       (const cplx*)h->partial, P, h->eig_small);
    KLAUNCH_CHECK();
  }
  HIPCHK(hipMemcpyAsync(out, h->eig_small, 4096 * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
  return stream_sync(h);
}

// buf_dst = buf_src Y (sub < 0) or buf_dst = buf_sub - buf_src Y, Y[64*64] row-major complex128 (dst != src)
int sw_eig_rotate(sw_engine* h, int src, const double* Y, int dst, int sub) {
  SWCHK(eig_check(h, src));
  SWCHK(eig_check(h, dst));
  if (sub >= 0) SWCHK(eig_check(h, sub));
  if (src == dst || !Y) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(h->eig_small, Y, 4096 * sizeof(cplx), hipMemcpyHostToDevice, h->stream));
  {
    LaunchScope ls(h, T_AXPY);
    hipLaunchKernelGGL(swk::k_block_rotate, dim3(std::min(2048, h->eig_n / SW_WAVES_PER_BLOCK)), dim3(SW_BLOCK),
                       0, h->stream, (const cplx*)h->eig_buf[src], (const cplx*)h->eig_small,
                       (const cplx*)(sub >= 0 ? h->eig_buf[sub] : nullptr), h->eig_buf[dst], h->eig_n);
    KLAUNCH_CHECK();
  }
  return stream_sync(h);
}

// The first k columns of buf_src as k host vectors of length n in the reference order.
int sw_eig_fetch(sw_engine* h, int src, int k, double* out) {
  SWCHK(eig_check(h, src));
  if (k < 1 || k > 64 || !out) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[h->eig_hid].lv[h->eig_level];
  return unpack_host(h, lv, k, h->eig_buf[src], out, 64);
}

// Current value of an engine switch (the counterpart of sw_set_option: what a caller saves before an A/B run
// and restores afterwards); "direct_fallbacks": how often a directly solved level missed the tolerance
// and the iterative path took over (read-only).
int sw_get_option(sw_engine* h, const char* name, double* value) {
  if (!h || !name || !value) return 1;
  struct Ent { const char* name; double v; };
  const Ent tab[] = {
      {"use_mfma", (double)h->use_mfma}, {"bsr_map", (double)h->bsr_map}, {"dense_map", (double)h->dense_map},
      {"bsr_sub", (double)h->bsr_sub}, {"bsr_stages", (double)h->bsr_stages},
      {"dense_stages", (double)h->dense_stages}, {"bsr_nt", (double)h->bsr_nt}, {"bsr_xreg", (double)h->bsr_xreg},
      {"p_even", (double)h->p_even}, {"ell_order", (double)h->ell_order}, {"bench_what", (double)h->bench_what},
      {"bench_mode", (double)h->bench_mode}, {"stencil_nt", (double)h->stencil_nt},
      {"stencil_spw", (double)h->stencil_spw}, {"stencil_tile", (double)h->stencil_tile}, {"cgs2", (double)h->cgs2},
      {"inner_cgs2", (double)h->inner_cgs2}, {"eo_direct", (double)h->eo_direct}, {"eo_solve", (double)h->eo_solve},
      {"pyth_last", (double)h->pyth_last}, {"precond_f32", (double)h->precond_f32},
      {"f32_tiles", (double)h->f32_tiles}, {"f32_splitk", (double)h->f32_splitk},
      {"dot_blocks", (double)g_dot_blocks}, {"f32_krylov", (double)h->f32_krylov},
      {"f32_pairs", (double)h->f32_pairs}, {"f32_stages", (double)h->f32_stages},
      {"f32_dense_stages", (double)h->f32_dense_stages}, {"stop_factor", h->stop_factor},
      {"fused_reduce", (double)h->fused_reduce}, {"eo_skew", (double)h->eo_skew},
      {"direct_small", (double)h->direct_small}, {"gram_cycle", (double)h->gram_cycle},
      {"lgmres_aug", (double)h->lgmres_aug}, {"verify", (double)h->verify}, {"lazy_sync", (double)h->lazy_sync},
      {"mfma_3m", (double)h->mfma_3m}, {"eo_skew_chunk", (double)h->eo_skew_chunk},
      {"gj_block", (double)h->gj_block}, {"eo_product", (double)h->eo_product},       {"eo_tile", (double)h->eo_tile}, {"dense_lds", (double)h->dense_lds},
      {"mfma3_tiles", (double)h->mfma3_tiles}, {"mfma_ops", (double)h->mfma_ops},
      {"mfma_small_tiles", (double)h->mfma_small_tiles}, {"mfma_tiles", (double)h->mfma_tiles},
      {"direct_fallbacks", (double)h->direct_fallbacks}, {"alloc_seconds", h->alloc_s},
      {"alloc_calls", (double)h->alloc_calls}, {"alloc_gbytes", h->alloc_bytes * 1e-9},
      {"pool_hits", (double)h->pool_hits},
      {"eo_tile_dbg", (double)h->eo_tile_dbg}};
  for (const Ent& e : tab)
    if (std::strcmp(name, e.name) == 0) {
      *value = e.v;
      return 0;
    }
  return sw_fail(h, "unknown option %s", name);
}

int sw_set_deflation(sw_engine* h, int k, const double* U) {
  SWCHK(check_hier(h, 0, 0, false));
  if (k < 0 || k > SW_MAX_DEFL) return sw_fail(h, "deflation rank %d out of [0,%d]", k, SW_MAX_DEFL);
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[0].lv[0];
  h->kd = 0;
  if (k == 0) return 0;
  if (!U) return sw_fail(h, "null deflation vectors");
  if (lv.n <= 0) return sw_fail(h, "level 0 undefined");
  const std::complex<double>* Uh = (const std::complex<double>*)U;
  std::vector<std::complex<double>> Ui((size_t)lv.n * k);
  for (int i = 0; i < lv.n; ++i) {
    const int r = lv.h_rowmap.empty() ? i : lv.h_rowmap[i];
    for (int q = 0; q < k; ++q) Ui[(size_t)r * k + q] = Uh[(size_t)i * k + q];
  }
  SWCHK(upload(h, (std::complex<double>**)&h->U, Ui.data(), Ui.size()));
  h->kd = k;
  return 0;
}

int sw_set_level_deflation(sw_engine* h, int level, int k, const double* V) {
  SWCHK(check_hier(h, 0, level, false));
  if (k < 0 || k > SW_MAX_DEFL) return sw_fail(h, "deflation rank %d out of [0,%d]", k, SW_MAX_DEFL);
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[0].lv[level];
  h->lkd[level] = 0;
  if (k == 0) return 0;
  if (!V) return sw_fail(h, "null deflation vectors");
  if (lv.n <= 0) return sw_fail(h, "level %d undefined", level);
  const std::complex<double>* Vh = (const std::complex<double>*)V;
  std::vector<std::complex<double>> Vi((size_t)lv.n * k);
  for (int i = 0; i < lv.n; ++i) {
    const int r = lv.h_rowmap.empty() ? i : lv.h_rowmap[i];
    for (int q = 0; q < k; ++q) Vi[(size_t)r * k + q] = Vh[(size_t)i * k + q];
  }
  SWCHK(upload(h, (std::complex<double>**)&h->lV[level], Vi.data(), Vi.size()));
  h->lkd[level] = k;
  return 0;
}

int sw_set_perm(sw_engine* h, int level, int64_t shift) {
  SWCHK(check_hier(h, 0, level, false));
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[0].lv[level];
  if (shift < 0) {
    SWCHK(dev_free(h, h->perm_src[level]));
    h->perm_src[level] = nullptr;
    return 0;
  }
  if (lv.n <= 0) return sw_fail(h, "level %d undefined", level);
  if (shift >= lv.n) return sw_fail(h, "shift %lld >= n", (long long)shift);
  // (Pperm^T v)[i] = v[(i - shift) mod n]   (multigrid.py:151-153)
  std::vector<int> src(lv.n);
  for (int i = 0; i < lv.n; ++i) {
    const int s = (int)(((int64_t)i - shift + lv.n) % lv.n);
    const int ri = lv.h_rowmap.empty() ? i : lv.h_rowmap[i];
    const int rs = lv.h_rowmap.empty() ? s : lv.h_rowmap[s];
    src[ri] = rs;
  }
  SWCHK(upload(h, &h->perm_src[level], src.data(), src.size()));
  return 0;
}

int sw_set_rhsmap(sw_engine* h, int level, int n, const int64_t* indptr, const int32_t* indices,
                  const double* data) {
  SWCHK(check_hier(h, 0, level, false));
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[0].lv[level];
  if (lv.n != n) return sw_fail(h, "rhs map size %d != level size %d", n, lv.n);
  SWCHK(free_op(h, h->rhsmap[level]));
  std::vector<int> rows_int;
  if (!lv.h_rowmap.empty()) {
    rows_int.resize(n);
    for (int i = 0; i < n; ++i) rows_int[lv.h_rowmap[i]] = i;
  }
  return build_ell(h, h->rhsmap[level], n, n, indptr, indices, (const std::complex<double>*)data,
                   rows_int, lv.h_rowmap, 1);
}

// ---- building blocks ---------------------------------------------------------------------
static int simple_op(sw_engine* h, int hid, int level, int nb, const double* X, double* Y, int what) {
  SWCHK(check_hier(h, hid, level, true));
  if (nb <= 0 || !X || !Y) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  const int nbp = pad64(nb);
  Level& lv = H.lv[level];
  cplx *a, *b;
  if (what == 0) {  // dirac
    if (level == H.nlevels - 1 && !lv.stencil && !lv.A.set)
      return sw_fail(h, "coarsest level operator was not supplied");
    SWCHK(io_vectors(h, lv, nbp, &a, &b));
    SWCHK(pack_host(h, lv, nb, X, a, nbp));
    SWCHK(apply_op(h, lv, 0, a, nullptr, b, nbp));
    return unpack_host(h, lv, nb, b, Y, nbp);
  }
  if (what == 3) {  // coarsest inverse
    Level& ll = H.lv[H.nlevels - 1];
    SWCHK(io_vectors(h, ll, nbp, &a, &b));
    SWCHK(pack_host(h, ll, nb, X, a, nbp));
    SWCHK(apply_coarsest(h, H, a, b, nbp));
    return unpack_host(h, ll, nb, b, Y, nbp);
  }
  if (level + 1 >= H.nlevels) return sw_fail(h, "no transfer at the coarsest level");
  Level& lc = H.lv[level + 1];
  cplx *c, *d;
  SWCHK(io_vectors(h, lv, nbp, &a, &b));
  SWCHK(io_vectors(h, lc, nbp, &c, &d));
  if (what == 1) {  // restrict
    SWCHK(pack_host(h, lv, nb, X, a, nbp));
    SWCHK(launch_ell(h, lv.R, 0, a, nullptr, c, nbp, T_R));
    return unpack_host(h, lc, nb, c, Y, nbp);
  }
  SWCHK(pack_host(h, lc, nb, X, c, nbp));
  SWCHK(launch_ell(h, lv.P, 0, c, nullptr, a, nbp, T_P));
  return unpack_host(h, lv, nb, a, Y, nbp);
}

int sw_apply_dirac(sw_engine* h, int hid, int level, int nb, const double* X, double* Y) {
  return simple_op(h, hid, level, nb, X, Y, 0);
}
int sw_restrict(sw_engine* h, int hid, int level, int nb, const double* X, double* Y) {
  return simple_op(h, hid, level, nb, X, Y, 1);
}
int sw_prolong(sw_engine* h, int hid, int level, int nb, const double* X, double* Y) {
  return simple_op(h, hid, level, nb, X, Y, 2);
}
int sw_coarsest(sw_engine* h, int hid, int nb, const double* X, double* Y) {
  return simple_op(h, hid, 0, nb, X, Y, 3);
}

int sw_vcycle(sw_engine* h, int hid, int level0, int nb, const double* B, double* X) {
  SWCHK(check_hier(h, hid, level0, true));
  if (nb <= 0 || !B || !X) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  const int nbp = pad64(nb);
  Level& lv = H.lv[level0];
  cplx *a, *b;
  SWCHK(io_vectors(h, lv, nbp, &a, &b));
  SWCHK(pack_host(h, lv, nb, B, a, nbp));
  if (h->precond_f32 && f32_capable(H, level0)) SWCHK(vcycle_f32_boundary(h, H, level0, a, b, nbp));
  else SWCHK(vcycle(h, H, level0, a, b, nbp));
  return unpack_host(h, lv, nb, b, X, nbp);
}

// ---------------------------------------------------------------------------------------------
// Outer solve of an even-odd smoothed stencil level on the EVEN-ODD REDUCED system.
//   A x = b  <=>  S x_e = b'_e,  b'_e = b_e - A_eo A_oo^-1 b_o,  x_o = A_oo^-1 (b_o - A_oe x_e),
//   S = A_ee - A_eo A_oo^-1 A_oe  (here A_oo = D, the Schur complement k_schur_step applies).
// The cycle's even-odd smoother leaves the odd residual exactly zero, so in the full-system FGMRES the
// odd halves of all Krylov vectors carry no information; here they do not exist: every vector of the
// Krylov solver (basis, directions, iterate, residual) is a HALF vector -- the first n / 2 rows of
// the parity-sorted level --, the operator is one k_schur_step<0> (two half passes instead of the
// stencil's four), and the preconditioner is the even block of the same multigrid cycle,
// M_S r_e = [M (r_e; 0)]_e ~ (A^-1)_ee = S^-1: restriction from the even columns only (Level::Re), no
// hop before the Schur steps (b_o = 0 makes b' = r_e) and none after them (the odd half is not
// needed).  The residual of the reduced system IS the residual of the full one (its odd half vanishes
// identically once x_o is set from x_e), and it is measured against ||b|| of the full system, so the
// stopping criterion is the reference's (multigrid.py:347-366).  Per iteration on the lattice level:
// about 47 half-vector passes instead of 64.
// ---------------------------------------------------------------------------------------------
static swk::StencilArgs eo_stencil_args(sw_engine* h, Level& lv, int nbp) {
  swk::StencilArgs a;
  a.L = lv.L;
  a.Vh = lv.L * lv.L / 2;
  a.diag = 4.0 + lv.mass;
  a.U1 = lv.U1;
  a.U2 = lv.U2;
  a.nbp = nbp;
  a.nt_store = 0;
  a.tile_w = lv.L;
  if (lv.L > 256) a.tile_w = (lv.L % 256 == 0) ? 256 : ((lv.L % 64 == 0) ? 64 : lv.L);
  if (h->stencil_tile > 0 && lv.L % h->stencil_tile == 0 && h->stencil_tile % 2 == 0) a.tile_w = h->stencil_tile;
  a.w = cplx{0.0, 0.0};
  return a;
}

// Y_e = S X_e (mode 0) or Bp_e - S X_e (mode 1); all three are half vectors
static int schur_apply(sw_engine* h, Level& lv, int mode, const cplx* X, const cplx* Bp, cplx* Y, int nbp) {
  swk::StencilArgs a = eo_stencil_args(h, lv, nbp);
  LaunchScope ls(h, T_SCHUR_OP);
  if (mode == 0) return launch_schur_step<0>(h, a, X, Bp, Y, nbp);
  return launch_schur_step<1>(h, a, X, Bp, Y, nbp);
}

static bool eo_solve_eligible(sw_engine* h, Hier& H, int level) {
  if (!h->eo_solve || level != 0 || H.nlevels < 2) return false;
  Level& lv = H.lv[0];
  if (!(lv.stencil && lv.rich && lv.gm_m == 0 && !lv.w_eo.empty() && lv.w_pre.empty() && lv.P.set &&
        lv.R.set && !h->cgs2 && h->p_even && (lv.n % 2 == 0)))
    return false;
  // single-precision preconditioner: only its default form (complex64 cycle AND complex64 Krylov basis)
  if (h->precond_f32 && !(h->f32_krylov && f32_capable(H, 0))) return false;
  if (ensure_even_orders(h, H) != 0) return false;
  return lv.Re.set && lv.P.order_even != nullptr;
}

// Xout_e = [M (Bin_e; 0)]_e : the even block of the level-0 cycle (see above); half vectors in and out
static int vcycle_even(sw_engine* h, Hier& H, const cplx* Bin, cplx* Xout, int nbp) {
  Level& lv = H.lv[0];
  Level& lc = H.lv[1];
  SWCHK(ensure_level_ws(h, lv, nbp));
  SWCHK(ensure_level_ws(h, lc, nbp));
  SWCHK(ensure_even_orders(h, H));
  if (!lv.Re.set) return sw_fail(h, "internal: even-column restrictor missing");
  SWCHK(launch_ell(h, lv.Re, 0, Bin, nullptr, lc.b, nbp, T_R));
  SWCHK(coarse_correction(h, H, 0, nbp));
  // (Xout is a HALF-length array: the prolongation must write the even sites only)
  if (!(lv.P.order_even && h->p_even))
    return sw_fail(h, "internal: even-odd reduced solve needs the even-sites-only prolongation (p_even)");
  if (h->eo_product && !lv.q_w.empty() && lv.q_w.size() + 1 == lv.w_eo.size()) {
    // product form: the coarse correction lands in Xout and is smoothed there; the two halves of lv.t are scratch
    SWCHK(launch_ell(h, lv.P, 0, lc.x, nullptr, Xout, nbp, T_P, cplx{0.0, 0.0}, true));
    return schur_product_steps(h, lv, Xout, lv.t, lv.t + (size_t)(lv.n / 2) * nbp, Bin, nbp);
  }
  // ping-pong between lv.t and Xout (only their even halves are touched) so that the last step lands in Xout
  const bool odd_steps = (lv.w_eo.size() & 1) != 0;
  cplx* cur = odd_steps ? lv.t : Xout;
  cplx* nxt = odd_steps ? Xout : lv.t;
  SWCHK(launch_ell(h, lv.P, 0, lc.x, nullptr, cur, nbp, T_P, cplx{0.0, 0.0}, true));
  cplx* res = nullptr;
  SWCHK(schur_steps(h, lv, cur, nxt, Bin, nbp, &res));
  if (res != Xout) return sw_fail(h, "internal: even-odd smoother ended in the wrong buffer");
  return 0;
}

// the same in complex64 (option precond_f32): half vectors of complex64 in and out
static int vcycle32_even(sw_engine* h, Hier& H, const cplxf* Bin, cplxf* Xout, int nbp) {
  Level& lv = H.lv[0];
  Level& lc = H.lv[1];
  const int last = H.nlevels - 1;
  SWCHK(ensure_level_ws32(h, lv, nbp));
  SWCHK(ensure_level_ws32(h, lc, nbp));
  SWCHK(ensure_even_orders(h, H));
  if (!lv.Re.set) return sw_fail(h, "internal: even-column restrictor missing");
  SWCHK(mirror_op32(h, lv.Re));
  SWCHK(launch_ell32(h, lv.Re, 0, Bin, nullptr, lc.b32, nbp, T_R));
  if (lv.kcycle > 0 && 1 < last) {
    // (as vcycle32: the few-step inner FGMRES of a K-cycle stays fp64, preconditioned in complex64)
    const size_t cc = (size_t)lc.n * nbp;
    SWCHK(ensure_level_ws(h, lc, nbp));
    SWCHK(cast_vec(h, (const cplxf*)lc.b32, lc.b, cc, T_AXPY));
    SWCHK(ensure_krylov(h, lc.kws, lv.kcycle, lc.n, nbp, false));
    SWCHK(fgmres(h, H, 1, lc.b, lc.x, 0.0, lv.kcycle, lv.kcycle, false, lc.kws, nbp, nullptr, true));
    SWCHK(cast_vec(h, (const cplx*)lc.x, lc.x32, cc, T_AXPY));
  } else {
    SWCHK(vcycle32(h, H, 1, lc.b32, lc.x32, nbp));
  }
  if (!(lv.P.order_even && h->p_even))
    return sw_fail(h, "internal: even-odd reduced solve needs the even-sites-only prolongation (p_even)");
  const bool odd_steps = (lv.w_eo.size() & 1) != 0;
  cplxf* start = odd_steps ? lv.t32 : Xout;
  cplxf* other = odd_steps ? Xout : lv.t32;
  SWCHK(launch_ell32(h, lv.P, 0, lc.x32, nullptr, start, nbp, T_P, cplxf{0.f, 0.f}, true));
  return eo_smooth32(h, lv, Bin, start, other, Xout, nbp, true);
}

// batched flexible GMRES(m) on the even-odd reduced system of the stencil level (one Gram-Schmidt
// pass, same scalar kernels, freezing, lazy read-back and true-residual verification as fgmres).
// With the single-precision preconditioner (precond_f32 + f32_krylov) the restart cycle is complex64 as
// in fgmres: basis, directions and w = S z in complex64 (S applied in single precision), inner products
// accumulated in fp64; residual b' - S x, iterate and convergence check fp64, once per restart.
static int fgmres_eo(sw_engine* h, Hier& H, const cplx* B, cplx* X, double tol, int maxiter, int m,
                     KrylovWS& ws, int nbp, int* iters_total) {
  Level& lv = H.lv[0];
  const int hid_idx = (int)(&H - &h->hier[0]);
  const int check_from = h->lazy_sync ? std::max(0, h->sync_hint[hid_idx][0] - 2) : 0;
  const int n2 = lv.n / 2;
  const size_t vec = (size_t)n2 * nbp;
  const int tb = 256, tg = (nbp + tb - 1) / tb;
  const double tol_stop = tol * h->stop_factor;
  SWCHK(ensure_level_ws(h, lv, nbp));
  SWCHK(reset_slots(h));
  swk::StencilArgs a = eo_stencil_args(h, lv, nbp);
  const int bpc = (a.Vh + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const dim3 grid(bpc * (nbp / 64));
  const double di = 1.0 / a.diag;
  cplx* bp = ws.xacc;               // b'_e
  const bool k32 = h->precond_f32 && h->f32_krylov && f32_capable(H, 0);
  if (k32) {
    SWCHK(ensure_f32(h, H));
    SWCHK(ensure_level_ws32(h, lv, nbp));
    const size_t full = (size_t)lv.n * nbp;      // (sized as fgmres sizes them: both may run on this handle)
    if (!ws.Z32) {
      SWCHK(dev_realloc(h, &ws.Z32, full * m));
      SWCHK(dev_realloc(h, &ws.v32, full));
    }
    if (!ws.V32) SWCHK(dev_realloc(h, &ws.V32, full * m));
  }
  {
    // ||b|| of the FULL system fixes normb (the reference's stopping criterion); b'_e = b_e + H_eo b_o / D
    PtrList pl;
    pl.p[0] = B;
    const swk::FgTail tbeg = tail_begin(ws, 1);
    SWCHK(multidot(h, pl, 1, B, lv.n, nbp, ws.nrm, nullptr, nullptr, &tbeg));
    LaunchScope ls(h, T_SCHUR);
    if (h->profiling) h->twork[T_SCHUR] += (double)a.Vh * (96.0 * nbp + 64.0);
    hipLaunchKernelGGL((swk::k_eo_hop<0>), grid, dim3(SW_BLOCK), 0, h->stream, B, B, bp, a, 1.0, di, bpc);
    KLAUNCH_CHECK();
  }
  // the iterate lives in the even half of X; its odd half is set once at the end
  {
    LaunchScope ls(h, T_AXPY);
    HIPCHK(hipMemsetAsync(X, 0, vec * sizeof(cplx), h->stream));
  }
  int done = 0;
  bool converged = false;
  const cplx* Rcur = bp;
  while (done < maxiter && !converged) {
    {
      PtrList pl;
      pl.p[0] = Rcur;
      const swk::FgTail tbeg = tail_begin(ws, 0);
      SWCHK(multidot(h, pl, 1, Rcur, n2, nbp, ws.nrm, nullptr, nullptr, &tbeg));
    }
    auto vt = [&](int k) -> const cplx* { return k == 0 ? Rcur : ws.V + vec * (k - 1); };
    auto vt32 = [&](int k) -> const cplxf* { return k == 0 ? ws.v32 : ws.V32 + vec * (k - 1); };
    if (k32) SWCHK(cast_vec(h, Rcur, ws.v32, vec, T_AXPY));
    int j = 0;
    const int jmax = std::min(m, maxiter - done);
    for (; j < jmax; ++j) {
      cplx* zj = ws.Z + vec * j;
      cplx* w = ws.V + vec * j;          // becomes vtilde_{j+1}
      const bool last = h->pyth_last && j == jmax - 1 && m <= 8;
      int* slot = nullptr;
      SWCHK(take_slot(h, &slot));
      const swk::FgTail th = tail_hess(ws, j, false, last, tol, tol_stop, done, slot);
      if (k32) {
        cplxf* zj32 = ws.Z32 + vec * j;
        cplxf* w32 = ws.V32 + vec * j;
        SWCHK(vcycle32_even(h, H, vt32(j), zj32, nbp));
        SWCHK(schur_apply32(h, lv, zj32, w32, nbp));
        swk::PtrListT<cplxf> pv32;
        for (int k = 0; k <= j; ++k) pv32.p[k] = vt32(k);
        pv32.p[j + 1] = w32;
        SWCHK(multidot(h, pv32, last ? j + 2 : j + 1, (const cplxf*)w32, n2, nbp, ws.h1, ws.sc.svec, ws.c1,
                       last ? &th : nullptr));
        if (!last)
          SWCHK(multiaxpy(h, pv32, j + 1, ws.c1, -1.0, (const cplxf*)w32, w32, n2, nbp, ws.nrm, nullptr, &th));
      } else {
      SWCHK(vcycle_even(h, H, vt(j), zj, nbp));
      SWCHK(schur_apply(h, lv, 0, zj, nullptr, w, nbp));
      PtrList pv;
      for (int k = 0; k <= j; ++k) pv.p[k] = vt(k);
      pv.p[j + 1] = w;
      SWCHK(multidot(h, pv, last ? j + 2 : j + 1, w, n2, nbp, ws.h1, ws.sc.svec, ws.c1, last ? &th : nullptr));
      if (!last) SWCHK(multiaxpy(h, pv, j + 1, ws.c1, -1.0, w, w, n2, nbp, ws.nrm, nullptr, &th));
      }
      if (done + j + 1 >= check_from || done + j + 1 >= maxiter || ((done + j + 1) & 7) == 0) {
        int left = 0;
        SWCHK(read_slot(h, slot, &left));
        if (left == 0) {
          converged = true;
          ++j;
          break;
        }
      }
    }
    const int k = j;
    {
      LaunchScope ls(h, T_OTHER);
      hipLaunchKernelGGL(swk::k_fg_solve, dim3(tg), dim3(tb), 0, h->stream, ws.sc, k);
      KLAUNCH_CHECK();
    }
    if (k32) {
      swk::PtrListT<cplxf> pz;
      for (int q = 0; q < k; ++q) pz.p[q] = ws.Z32 + vec * q;
      SWCHK(multiaxpy(h, pz, k, ws.sc.ys, 1.0, X, X, n2, nbp, nullptr));
    } else {
      PtrList pz;
      for (int q = 0; q < k; ++q) pz.p[q] = ws.Z + vec * q;
      SWCHK(multiaxpy(h, pz, k, ws.sc.ys, 1.0, X, X, n2, nbp, nullptr));
    }
    done += k;
    if (converged && h->verify) {
      // true residual of the reduced system = true residual of the full one
      SWCHK(schur_apply(h, lv, 1, X, bp, ws.rres, nbp));
      PtrList pr;
      pr.p[0] = ws.rres;
      int* slot = nullptr;
      SWCHK(take_slot(h, &slot));
      const swk::FgTail tv = tail_verify(ws, tol, tol_stop, slot);
      SWCHK(multidot(h, pr, 1, ws.rres, n2, nbp, ws.nrm, nullptr, nullptr, &tv));
      int left = 0;
      SWCHK(read_slot(h, slot, &left));
      if (left != 0) {
        converged = false;
        Rcur = ws.rres;
        continue;
      }
    } else if (!converged && done < maxiter) {
      SWCHK(schur_apply(h, lv, 1, X, bp, ws.rres, nbp));
      Rcur = ws.rres;
    }
  }
  {
    // x_o = (b_o + H_oe x_e) / D
    LaunchScope ls(h, T_SCHUR);
    if (h->profiling) h->twork[T_SCHUR] += (double)a.Vh * (96.0 * nbp + 64.0);
    hipLaunchKernelGGL((swk::k_eo_hop<1>), grid, dim3(SW_BLOCK), 0, h->stream, B, (const cplx*)X, X, a, di, di,
                       bpc);
    KLAUNCH_CHECK();
  }
  if (iters_total) *iters_total = done;
  h->sync_hint[hid_idx][0] = converged ? done : 0;
  return 0;
}

// The same solve with restart cycles in GRAM-MATRIX form (option gram_cycle, default).  A cycle of L <= m
// steps builds z_0 = M r, w_0 = S z_0, z_1 = M w_0, w_1 = S z_1, ... with NO inner products or
// orthogonalisation in between -- the same Krylov space FGMRES(m) spans --, then one pass over [r, w_0 ..
// w_{L-1}] yields all their inner products and the minimal-residual combination follows per probe from a
// Cholesky-factorised L x L system (fg_gram_col, which also gives the residual norm of every nested
// sub-cycle: per-probe iteration counts as before).  Per cycle of three: 12 half-vector passes of BLAS-1 and
// residual instead of 29, 4 launches instead of 9.  Every cycle ends with the TRUE residual b' - S x, so the
// convergence test always sees true residuals; the read-backs are per cycle, not per iteration: cycles run
// unobserved until the previous batch's iteration count is within one cycle, then the residuals of all
// probes come back (a few KB) and the last cycle is cut to the length they call for.
static int fgmres_eo_gram(sw_engine* h, Hier& H, const cplx* B, cplx* X, double tol, int maxiter, int m,
                          KrylovWS& ws, int nbp, int* iters_total) {
  Level& lv = H.lv[0];
  const int hid_idx = (int)(&H - &h->hier[0]);
  const int hint = h->lazy_sync ? h->sync_hint[hid_idx][0] : 0;
  const int n2 = lv.n / 2;
  const size_t vec = (size_t)n2 * nbp;
  const double tol_stop = tol * h->stop_factor;
  SWCHK(ensure_level_ws(h, lv, nbp));
  SWCHK(reset_slots(h));
  swk::StencilArgs a = eo_stencil_args(h, lv, nbp);
  const int bpc = (a.Vh + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK;
  const dim3 grid(bpc * (nbp / 64));
  const double di = 1.0 / a.diag;
  cplx* bp = ws.xacc;               // b'_e
  {
    // ||b|| of the FULL system fixes normb (the reference's stopping criterion); b'_e = b_e + H_eo b_o / D
    PtrList pl;
    pl.p[0] = B;
    const swk::FgTail tbeg = tail_begin(ws, 1);
    SWCHK(multidot(h, pl, 1, B, lv.n, nbp, ws.nrm, nullptr, nullptr, &tbeg));
    LaunchScope ls(h, T_SCHUR);
    if (h->profiling) h->twork[T_SCHUR] += (double)a.Vh * (96.0 * nbp + 64.0);
    hipLaunchKernelGGL((swk::k_eo_hop<0>), grid, dim3(SW_BLOCK), 0, h->stream, B, B, bp, a, 1.0, di, bpc);
    KLAUNCH_CHECK();
  }
  {
    LaunchScope ls(h, T_AXPY);
    HIPCHK(hipMemsetAsync(X, 0, vec * sizeof(cplx), h->stream));
  }
  std::vector<std::complex<double>> hrr(nbp), hrr0(nbp);
  int done = 0;
  bool converged = false;
  const cplx* Rcur = bp;
  int L = std::min(m, maxiter);
  while (done < maxiter && !converged) {
    L = std::max(1, std::min(L, std::min(m, maxiter - done)));
    for (int j = 0; j < L; ++j) {
      const cplx* vin = (j == 0) ? Rcur : ws.V + vec * (j - 1);
      SWCHK(vcycle_even(h, H, vin, ws.Z + vec * j, nbp));
      SWCHK(schur_apply(h, lv, 0, ws.Z + vec * j, nullptr, ws.V + vec * j, nbp));
    }
    int* slot = nullptr;
    SWCHK(take_slot(h, &slot));
    {
      PtrList pu;
      pu.p[0] = Rcur;
      for (int j = 0; j < L; ++j) pu.p[j + 1] = ws.V + vec * j;
      swk::FgTail tg{};
      tg.kind = SW_TAIL_GRAM;
      tg.s = ws.sc;
      tg.j = L;
      tg.h1 = ws.gram;
      tg.tol = tol;
      tg.tol_stop = tol_stop;
      tg.iter_base = done;
      tg.notconv = slot;
      SWCHK(multigram(h, pu, L + 1, n2, nbp, ws.gram, &tg));
    }
    {
      PtrList pz;
      for (int q = 0; q < L; ++q) pz.p[q] = ws.Z + vec * q;
      SWCHK(multiaxpy(h, pz, L, ws.sc.ys, 1.0, X, X, n2, nbp, nullptr));
    }
    done += L;
    // the true residual: input of the next cycle and of the final check
    SWCHK(schur_apply(h, lv, 1, X, bp, ws.rres, nbp));
    Rcur = ws.rres;
    // Observe?  Not while the previous batch's count says more than one further cycle is due.
    const bool observe = tol_stop > 0.0 && (hint <= 0 || done + m >= hint || done >= maxiter);
    int Lnext = m;
    if (observe) {
      int left = 0;
      SWCHK(read_slot(h, slot, &left));
      if (left == 0) {
        converged = true;
      } else {
        // how many more steps do the slowest probes need at the rate this cycle achieved?
        HIPCHK(hipMemcpy(hrr.data(), ws.sc.relres, sizeof(cplx) * nbp, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hrr0.data(), ws.sc.g, sizeof(cplx) * nbp, hipMemcpyDeviceToHost));
        double need = 1.0;
        for (int c = 0; c < nbp; ++c) {
          const double r1 = hrr[c].real(), r0 = hrr0[c].real();
          if (!(r1 >= tol_stop) || !(r0 > 0.0)) continue;
          const double rate = std::pow(std::min(0.9, std::max(1e-6, r1 / r0)), 1.0 / L);   // per step
          need = std::max(need, std::ceil(std::log(0.9 * tol_stop / r1) / std::log(rate)));
        }
        Lnext = (int)std::min<double>(m, need);
      }
    }
    if (converged && h->verify) {
      // (the decision above used the cycle's own estimate of |r_L|; the residual just formed is the true one)
      PtrList pr;
      pr.p[0] = ws.rres;
      int* vslot = nullptr;
      SWCHK(take_slot(h, &vslot));
      const swk::FgTail tv = tail_verify(ws, tol, tol_stop, vslot);
      SWCHK(multidot(h, pr, 1, ws.rres, n2, nbp, ws.nrm, nullptr, nullptr, &tv));
      int left = 0;
      SWCHK(read_slot(h, vslot, &left));
      if (left != 0) {
        converged = false;
        Lnext = 1;
      }
    }
    L = Lnext;
  }
  {
    // x_o = (b_o + H_oe x_e) / D
    LaunchScope ls(h, T_SCHUR);
    if (h->profiling) h->twork[T_SCHUR] += (double)a.Vh * (96.0 * nbp + 64.0);
    hipLaunchKernelGGL((swk::k_eo_hop<1>), grid, dim3(SW_BLOCK), 0, h->stream, B, (const cplx*)X, X, a, di, di,
                       bpc);
    KLAUNCH_CHECK();
  }
  if (iters_total) *iters_total = done;
  h->sync_hint[hid_idx][0] = converged ? done : 0;
  return 0;
}

static bool level_is_direct(sw_engine* h, Hier& H, int level) {
  return h->direct_small && level < H.nlevels - 1 && !H.lv[level].stencil && H.lv[level].dinv.set &&
         h->use_mfma;
}

// device-resident solve used by sw_solve and the probe drivers
static int solve_dev(sw_engine* h, int hid, int level0, const cplx* B, cplx* X, double tol,
                     int maxiter, int nbp, int* total) {
  Hier& H = h->hier[hid];
  Level& lv = H.lv[level0];
  if (level0 == H.nlevels - 1 && H.nlevels > 1) {
    // coarsest level: the dense inverse is the solve (multigrid.py:413-416)
    SWCHK(apply_coarsest(h, H, B, X, nbp));
    if (total) *total = 1;
    return 0;
  }
  if (level_is_direct(h, H, level0)) {
    // x = A^-1 b ; x += A^-1 (b - A x): dense inverse on the matrix cores, one refinement step
    SWCHK(ensure_level_ws(h, lv, nbp));
    SWCHK(launch_bsr(h, lv.dinv, 0, B, nullptr, X, nbp, T_COARSEST, cplx{0.0, 0.0}));
    SWCHK(apply_op(h, lv, 1, X, B, lv.r, nbp));
    SWCHK(launch_bsr(h, lv.dinv, 0, lv.r, nullptr, lv.t, nbp, T_COARSEST, cplx{0.0, 0.0}));
    SWCHK(vec_add(h, X, lv.t, X, lv.n, nbp));
    if (total) *total = 1;
    // The refinement step takes the inverse's own residual (eps * cond) below the solver tolerance on the
    // operators this was built for; that is MEASURED here, not assumed: ||b - A x|| / ||b|| per right-hand
    // side, further refinement steps while any of them is above stop_factor * tol, and the multigrid-
    // preconditioned FGMRES of the general path if three more steps do not get there.
    SWCHK(ensure_small(h, nbp));
    h->direct_relres.assign(nbp, 0.0);
    for (int pass = 0;; ++pass) {
      SWCHK(apply_op(h, lv, 1, X, B, lv.r, nbp));
      SWCHK(dot_into(h, lv.r, lv.r, lv.n, nbp, h->small));
      SWCHK(dot_into(h, B, B, lv.n, nbp, h->small + nbp));
      SWCHK(stream_sync(h));
      std::vector<std::complex<double>> nn(2 * (size_t)nbp);
      HIPCHK(hipMemcpy(nn.data(), h->small, sizeof(cplx) * 2 * nbp, hipMemcpyDeviceToHost));
      double worst = 0.0;
      for (int j = 0; j < nbp; ++j) {
        const double b2 = nn[nbp + j].real(), r2 = nn[j].real();
        const double rr = b2 > 0.0 ? std::sqrt(r2 / b2) : 0.0;
        h->direct_relres[j] = rr;
        worst = std::max(worst, rr);
      }
      if (!(worst > tol * h->stop_factor)) return 0;
      if (pass == 3) break;
      SWCHK(launch_bsr(h, lv.dinv, 0, lv.r, nullptr, lv.t, nbp, T_COARSEST, cplx{0.0, 0.0}));
      SWCHK(vec_add(h, X, lv.t, X, lv.n, nbp));
    }
    h->direct_relres.clear();      // the iterative path reports its own residuals
    h->direct_fallbacks++;
    const int mfb = std::min(h->restart, std::max(1, maxiter));
    SWCHK(ensure_krylov(h, lv.sws, mfb, lv.n, nbp, true));
    return fgmres(h, H, level0, B, X, tol, maxiter, mfb, true, lv.sws, nbp, total);
  }
  const int m = std::min(h->restart, std::max(1, maxiter));
  SWCHK(ensure_krylov(h, lv.sws, m, lv.n, nbp, true));
  if (eo_solve_eligible(h, H, level0)) {
    const bool k32 = h->precond_f32 && h->f32_krylov && f32_capable(H, 0);
    int P = 0, rpb = 0;
    row_blocking(lv.n / 2, nbp, true, &P, &rpb);
    if (h->gram_cycle && !k32 && m <= SW_GRAM_MAXL && fused_ok(h, P, nbp))
      return fgmres_eo_gram(h, H, B, X, tol, maxiter, m, lv.sws, nbp, total);
    return fgmres_eo(h, H, B, X, tol, maxiter, m, lv.sws, nbp, total);
  }
  return fgmres(h, H, level0, B, X, tol, maxiter, m, true, lv.sws, nbp, total);
}

static int fetch_iters(sw_engine* h, KrylovWS& ws, int nb, int32_t* iters, double* relres,
                       int fallback_iters) {
  std::vector<int> it(ws.nbp);
  std::vector<std::complex<double>> rr(ws.nbp);
  HIPCHK(hipMemcpy(it.data(), ws.sc.iters, sizeof(int) * ws.nbp, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rr.data(), ws.sc.relres, sizeof(cplx) * ws.nbp, hipMemcpyDeviceToHost));
  for (int j = 0; j < nb; ++j) {
    if (iters) iters[j] = it[j] >= 0 ? it[j] : fallback_iters;
    if (relres) relres[j] = rr[j].real();
  }
  return 0;
}

int sw_solve(sw_engine* h, int hid, int level0, int nb, const double* B, double* X, double tol,
             int maxiter, int32_t* iters, double* relres) {
  SWCHK(check_hier(h, hid, level0, true));
  if (nb <= 0 || !B || !X || maxiter < 1) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Hier& H = h->hier[hid];
  const int nbp = pad64(nb);
  Level& lv = H.lv[level0];
  cplx *a, *b;
  SWCHK(io_vectors(h, lv, nbp, &a, &b));
  SWCHK(pack_host(h, lv, nb, B, a, nbp));
  int total = 0;
  SWCHK(solve_dev(h, hid, level0, a, b, tol, maxiter, nbp, &total));
  SWCHK(unpack_host(h, lv, nb, b, X, nbp));
  if (level0 == H.nlevels - 1 && H.nlevels > 1) {
    for (int j = 0; j < nb; ++j) {
      if (iters) iters[j] = 1;
      if (relres) relres[j] = 0.0;
    }
    return 0;
  }
  if (level_is_direct(h, H, level0) && (int)h->direct_relres.size() >= nb) {
    // directly solved level: iteration count 1 as for the coarsest level, the MEASURED residual
    for (int j = 0; j < nb; ++j) {
      if (iters) iters[j] = 1;
      if (relres) relres[j] = h->direct_relres[j];
    }
    return 0;
  }
  return fetch_iters(h, lv.sws, nb, iters, relres, total);
}

// ---- probe batches ---------------------------------------------------------------------
static int ensure_probe_ws(sw_engine* h, int nbp) {
  if (h->pb_ws_nbp == nbp && h->pb_x0) return 0;
  Hier& H0 = h->hier[0];
  int nmax = 0;
  for (int l = 0; l < H0.nlevels; ++l) nmax = std::max(nmax, H0.lv[l].n);
  const size_t cnt = (size_t)nmax * nbp;
  SWCHK(dev_realloc(h, &h->pb_x0, cnt));
  SWCHK(dev_realloc(h, &h->pb_rhs, cnt));
  SWCHK(dev_realloc(h, &h->pb_z, cnt));
  SWCHK(dev_realloc(h, &h->pb_xc, cnt));
  SWCHK(dev_realloc(h, &h->pb_xc2, cnt));
  SWCHK(dev_realloc(h, &h->pb_y, cnt));
  SWCHK(dev_realloc(h, &h->pb_w, cnt));
  SWCHK(dev_realloc(h, &h->pb_w2, cnt));
  SWCHK(dev_realloc(h, &h->pb_xd, cnt));
  SWCHK(dev_realloc(h, &h->pb_est, (size_t)4 * nbp));
  SWCHK(dev_realloc(h, &h->pb_iters, (size_t)2 * nbp));
  h->pb_ws_nbp = nbp;
  return 0;
}

int sw_probes_upload_slot(sw_engine* h, int slot, int level, int nb, const int8_t* probes) {
  SWCHK(check_hier(h, 0, level, true));
  if (nb <= 0 || !probes) return sw_fail(h, "bad arguments");
  if (slot < 0 || slot >= 4096) return sw_fail(h, "probe slot %d out of range", slot);
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[0].lv[level];
  const size_t bytes = (size_t)nb * lv.n;
  if ((int)h->slots.size() <= slot) h->slots.resize(slot + 1);
  sw_engine::ProbeSlot& sl = h->slots[slot];
  if (sl.pending) {
    HIPCHK(hipStreamSynchronize(h->gen_stream));
    sl.pending = false;
  }
  if (sl.bytes < bytes) {
    SWCHK(dev_free(h, sl.p));
    sl.p = nullptr;
    void* q;
    SWCHK(dev_alloc(h, &q, bytes));
    sl.p = (int8_t*)q;
    sl.bytes = bytes;
  }
  HIPCHK(hipMemcpyAsync(sl.p, probes, bytes, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  sl.level = level;
  sl.nb = nb;
  return 0;
}

int sw_probes_select(sw_engine* h, int slot) {
  if (!h) return 1;
  if (slot < 0 || slot >= (int)h->slots.size() || !h->slots[slot].p)
    return sw_fail(h, "probe slot %d is empty", slot);
  h->pb_slot = slot;
  h->pb_probes = h->slots[slot].p;
  h->pb_level = h->slots[slot].level;
  h->pb_nb = h->slots[slot].nb;
  h->pb_nbp = pad64(h->pb_nb);
  return 0;
}

int sw_probes_upload(sw_engine* h, int level, int nb, const int8_t* probes) {
  SWCHK(sw_probes_upload_slot(h, 0, level, nb, probes));
  return sw_probes_select(h, 0);
}

// ---- device-side probe generation --------------------------------------------------------
int sw_probes_stream_set(sw_engine* h, const uint32_t* window) {
  if (!h) return 1;
  if (!window) return sw_fail(h, "null MT19937 window");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->gen_stream));      // a queued generation may still read the old window
  if (!h->mt_win0) SWCHK(dev_realloc(h, &h->mt_win0, (size_t)SW_MT_N));
  if (!h->mt_win) SWCHK(dev_realloc(h, &h->mt_win, (size_t)2 * SW_MT_N));
  if (!h->mt_poly) SWCHK(dev_realloc(h, &h->mt_poly, (size_t)SW_MT_N));
  HIPCHK(hipMemcpyAsync(h->mt_win0, window, SW_MT_N * sizeof(uint32_t), hipMemcpyHostToDevice,
                        h->stream));
  HIPCHK(hipMemcpyAsync(h->mt_win, window, SW_MT_N * sizeof(uint32_t), hipMemcpyHostToDevice,
                        h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->mt_cur = 0;
  h->mt_pos = 0;
  h->mt_dist = 0;
  h->mt_set = true;
  return 0;
}

// probes per generator segment: segments of ~64K draws (105 state blocks), whole probes
static inline int mt_probes_per_segment(int n) { return std::max(1, 65536 / n); }

int sw_probes_generate(sw_engine* h, int slot, int level, int nb, int kind, uint64_t pos) {
  SWCHK(check_hier(h, 0, level, true));
  if (!h->mt_set) return sw_fail(h, "no probe stream set (sw_probes_stream_set)");
  if (nb <= 0) return sw_fail(h, "bad arguments");
  if (kind != SW_PROBES_Z2 && kind != SW_PROBES_Z4) return sw_fail(h, "unknown probe kind %d", kind);
  if (slot < 0 || slot >= 4096) return sw_fail(h, "probe slot %d out of range", slot);
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[0].lv[level];
  const int n = lv.n;
  if (n % 4) return sw_fail(h, "device probe generation needs n %% 4 == 0 (n = %d)", n);
  const uint64_t total = (uint64_t)nb * (uint64_t)n;
  // Asynchronous, on the generation stream: the call returns once the work is queued, the slot's `ready`
  // event orders it ahead of the sw_hutch_run that consumes the slot.  (While profiling, generation
  // stays on the solve stream so that the per-launch event pairs bracket it.)
  // (SW_SINGLE_STREAM=1 keeps it on the solve stream as well: for tools that want one queue per process)
  static const bool single_stream = [] {
    const char* e = getenv("SW_SINGLE_STREAM");
    return e && e[0] == '1';
  }();
  hipStream_t gs = (h->profiling || single_stream) ? h->stream : h->gen_stream;
  // 1. move the resident window to `pos`
  if (pos < h->mt_pos) {
    HIPCHK(hipMemcpyAsync(h->mt_win + (size_t)SW_MT_N * h->mt_cur, h->mt_win0,
                          SW_MT_N * sizeof(uint32_t), hipMemcpyDeviceToDevice, gs));
    h->mt_pos = 0;
  }
  if (pos > h->mt_pos) {
    const uint64_t dist = pos - h->mt_pos;
    if (dist != h->mt_dist) {
      uint32_t poly[SW_MT_N];
      if (sw_mt_jump_poly(dist, poly) != 0) return sw_fail(h, "jump polynomial construction failed");
      // the previous polynomial may still be in use by a queued k_mt_jump: stream order protects it
      HIPCHK(hipMemcpyAsync(h->mt_poly, poly, sizeof poly, hipMemcpyHostToDevice, gs));
      HIPCHK(hipStreamSynchronize(gs));   // `poly` is a stack buffer
      h->mt_dist = dist;
    }
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_mt_jump, dim3(1), dim3(SW_MT_BLOCK), 0, gs,
                       (const uint32_t*)(h->mt_win + (size_t)SW_MT_N * h->mt_cur),
                       (const uint32_t*)h->mt_poly, h->mt_win + (size_t)SW_MT_N * (1 - h->mt_cur));
    KLAUNCH_CHECK();
    h->mt_cur = 1 - h->mt_cur;
    h->mt_pos = pos;
  }
  // 2. the polynomial family of the segment starts
  const int pps = mt_probes_per_segment(n);
  const uint64_t segdraws = (uint64_t)pps * (uint64_t)n;
  const int S = (nb + pps - 1) / pps;
  if (S > 1 && (h->mt_fam_seg != segdraws || h->mt_fam_count < S - 1)) {
    std::vector<uint32_t> fam((size_t)(S - 1) * SW_MT_N);
    for (int s = 1; s < S; ++s)
      if (sw_mt_jump_poly((uint64_t)s * segdraws, &fam[(size_t)(s - 1) * SW_MT_N]) != 0)
        return sw_fail(h, "jump polynomial construction failed");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->gen_stream));
    SWCHK(upload(h, &h->mt_family, fam.data(), fam.size()));
    h->mt_fam_seg = segdraws;
    h->mt_fam_count = S - 1;
  }
  // 3. generate into the slot
  if ((int)h->slots.size() <= slot) h->slots.resize(slot + 1);
  sw_engine::ProbeSlot& sl = h->slots[slot];
  if (sl.bytes < total) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->gen_stream));
    SWCHK(dev_free(h, sl.p));
    sl.p = nullptr;
    void* q;
    SWCHK(dev_alloc(h, &q, total));
    sl.p = (int8_t*)q;
    sl.bytes = total;
  }
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_mt_generate, dim3(S), dim3(SW_MT_BLOCK), 0, gs,
                       (const uint32_t*)(h->mt_win + (size_t)SW_MT_N * h->mt_cur),
                       (const uint32_t*)h->mt_family, (unsigned long long)segdraws,
                       (unsigned long long)total, kind, sl.p);
    KLAUNCH_CHECK();
  }
  sl.level = level;
  sl.nb = nb;
  if (!sl.ready) HIPCHK(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
  HIPCHK(hipEventRecord(sl.ready, gs));
  sl.pending = true;
  return 0;
}

int sw_probes_fetch(sw_engine* h, int slot, int8_t* out) {
  if (!h) return 1;
  if (!out) return sw_fail(h, "null output");
  if (slot < 0 || slot >= (int)h->slots.size() || !h->slots[slot].p)
    return sw_fail(h, "probe slot %d is empty", slot);
  HIPCHK(hipSetDevice(h->device));
  SWCHK(stream_sync(h));
  HIPCHK(hipStreamSynchronize(h->gen_stream));
  const sw_engine::ProbeSlot& sl = h->slots[slot];
  const size_t bytes = (size_t)sl.nb * h->hier[0].lv[sl.level].n;
  HIPCHK(hipMemcpy(out, sl.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

static int dot_into(sw_engine* h, const cplx* A, const cplx* Bv, int n, int nbp, cplx* out) {
  PtrList pl;
  pl.p[0] = A;
  return multidot(h, pl, 1, Bv, n, nbp, out);
}

static int record_iters(sw_engine* h, KrylovWS* ws, int total_or_const, std::vector<int32_t>& dst,
                        int nb) {
  dst.assign(nb, total_or_const);
  if (ws) {
    std::vector<int> it(ws->nbp);
    HIPCHK(hipMemcpy(it.data(), ws->sc.iters, sizeof(int) * ws->nbp, hipMemcpyDeviceToHost));
    for (int j = 0; j < nb; ++j) dst[j] = it[j] >= 0 ? it[j] : total_or_const;
  }
  return 0;
}

// out[r] = X[s] - sum_k U[s][k] (U^H X)[k],  s = srcrow[r] (NULL: identity)   (utils.py:221-225)
static int deflate(sw_engine* h, const cplx* U, int kd, const int* srcrow, const cplx* X, cplx* out,
                   int n, int nbp) {
  int P, rpb;
  row_blocking(n, nbp, true, &P, &rpb);
  cplx* cbuf = h->small + 8 * nbp;  // [kd][nbp], kd <= SW_MAX_DEFL
  for (int k0 = 0; k0 < kd; k0 += 32) {
    const int kc = std::min(32, kd - k0);
    SWCHK(ensure_partial(h, (size_t)P * kc * nbp * sizeof(cplx)));
    {
      LaunchScope ls(h, T_DEFL);
      dim3 grid(P, nbp / 64);
      if (kc <= 8)
        hipLaunchKernelGGL((swk::k_defl_dots<8>), grid, dim3(SW_BLOCK), 0, h->stream,
                           (const cplx*)(U + k0), kd, kc, X, n, nbp, rpb, h->partial);
      else if (kc <= 16)
        hipLaunchKernelGGL((swk::k_defl_dots<16>), grid, dim3(SW_BLOCK), 0, h->stream,
                           (const cplx*)(U + k0), kd, kc, X, n, nbp, rpb, h->partial);
      else
        hipLaunchKernelGGL((swk::k_defl_dots<32>), grid, dim3(SW_BLOCK), 0, h->stream,
                           (const cplx*)(U + k0), kd, kc, X, n, nbp, rpb, h->partial);
      KLAUNCH_CHECK();
    }
    {
      LaunchScope ls(h, T_DEFL);
      hipLaunchKernelGGL(swk::k_reduce_partials, dim3(kc, nbp / 64), dim3(SW_BLOCK), 0, h->stream,
                         h->partial, P, kc, nbp, cbuf + (size_t)k0 * nbp, (const cplx*)nullptr,
                         (cplx*)nullptr);
      KLAUNCH_CHECK();
    }
  }
  {
    LaunchScope ls(h, T_DEFL);
    dim3 grid((n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, nbp / 64);
    hipLaunchKernelGGL(swk::k_defl_apply, grid, dim3(SW_BLOCK), 0, h->stream, U, kd, cbuf, srcrow, X,
                       out, n, nbp);
    KLAUNCH_CHECK();
  }
  return 0;
}

int sw_hutch_run(sw_engine* h, int mode, int level, double tol, int maxiter) {
  SWCHK(check_hier(h, 0, level, true));
  if (h->pb_level != level || h->pb_nb <= 0) return sw_fail(h, "no probes uploaded for level %d", level);
  if (maxiter < 1) return sw_fail(h, "maxiter must be >= 1");
  HIPCHK(hipSetDevice(h->device));
  Hier& H0 = h->hier[0];
  const int nb = h->pb_nb, nbp = h->pb_nbp;
  Level& lv = H0.lv[level];
  const int n = lv.n;
  SWCHK(ensure_probe_ws(h, nbp));
  SWCHK(ensure_small(h, nbp));
  if (h->pb_slot >= 0 && h->pb_slot < (int)h->slots.size() && h->slots[h->pb_slot].pending) {
    // the slot was (or is being) generated on the generation stream
    HIPCHK(hipStreamWaitEvent(h->stream, h->slots[h->pb_slot].ready, 0));
    h->slots[h->pb_slot].pending = false;
  }
  // x0 <- probes
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_pack_i8, dim3((n + 63) / 64, nbp / 64), dim3(SW_BLOCK), 0, h->stream,
                       h->pb_probes, nb, n, (const int*)lv.rowmap, h->pb_x0, nbp);
    KLAUNCH_CHECK();
  }
  const int fine_hid = (level == 0 && h->hier[h->solver_hid].ready) ? h->solver_hid : 0;
  if (fine_hid != 0 && h->hier[fine_hid].lv[0].n != n)
    return sw_fail(h, "solver hierarchy level-0 size mismatch");
  if (mode == SW_MODE_HUTCHINSON) {
    if (level != 0) return sw_fail(h, "Hutchinson mode runs at level 0");
    // rhs = Pperm^T (x - U U^H x)          utils.py:221-233
    const int kd = h->kd;
    if (kd > 0) {
      SWCHK(deflate(h, h->U, kd, (const int*)h->perm_src[0], h->pb_x0, h->pb_rhs, n, nbp));
    } else {
      LaunchScope ls(h, T_DEFL);
      dim3 grid((n + SW_WAVES_PER_BLOCK - 1) / SW_WAVES_PER_BLOCK, nbp / 64);
      hipLaunchKernelGGL(swk::k_gather_rows, grid, dim3(SW_BLOCK), 0, h->stream,
                         (const int*)h->perm_src[0], h->pb_x0, h->pb_rhs, n, nbp);
      KLAUNCH_CHECK();
    }
    int total = 0;
    SWCHK(solve_dev(h, fine_hid, 0, h->pb_rhs, h->pb_z, tol, maxiter, nbp, &total));
    SWCHK(dot_into(h, h->pb_x0, h->pb_z, n, nbp, h->pb_est));   // e = x^H z   utils.py:249
    SWCHK(stream_sync(h));
    SWCHK(record_iters(h, &h->hier[fine_hid].lv[0].sws, total, h->last_iters_f, nb));
    h->last_iters_c.assign(nb, 0);
    return 0;
  }
  if (mode == SW_MODE_LEVEL) {
    // e = x0^H A_l^-1 (Bblock_perm_l Pperm_l^T) x0: the plain Hutchinson estimator of one level's
    // trace term (stochastic form of the coarsest-level term, stoch_trace.py:428-437)
    const cplx* xdef = h->pb_x0;
    if (h->rhsmap[level].set) {
      SWCHK(launch_ell(h, h->rhsmap[level], 0, xdef, nullptr, h->pb_rhs, nbp, T_OTHER));
      xdef = h->pb_rhs;
    }
    int total = 0;
    SWCHK(solve_dev(h, fine_hid, level, xdef, h->pb_z, tol, maxiter, nbp, &total));
    SWCHK(dot_into(h, h->pb_x0, h->pb_z, n, nbp, h->pb_est));
    SWCHK(stream_sync(h));
    if ((level == H0.nlevels - 1 && H0.nlevels > 1) || level_is_direct(h, h->hier[fine_hid], level))
      h->last_iters_f.assign(nb, 1);
    else SWCHK(record_iters(h, &h->hier[fine_hid].lv[level].sws, total, h->last_iters_f, nb));
    h->last_iters_c.assign(nb, 0);
    return 0;
  }
  if (mode != SW_MODE_MLMC && mode != SW_MODE_MLMC_SKIP) return sw_fail(h, "unknown mode %d", mode);
  const bool skip = (mode == SW_MODE_MLMC_SKIP);
  if (skip && level != 0) return sw_fail(h, "level skipping is defined for level 0 only");
  const int lcoarse = level + (skip ? 2 : 1);
  if (lcoarse >= H0.nlevels) return sw_fail(h, "no coarse level %d", lcoarse);
  // x_def = Bblock_perm * Pperm^T * x0      utils.py:288-290
  const cplx* xdef = h->pb_x0;
  if (h->lkd[level] > 0) {
    // x_def = x0 - V V^H x0                 utils.py:260-266 (defl_type exact / inexact_01)
    SWCHK(deflate(h, h->lV[level], h->lkd[level], nullptr, h->pb_x0, h->pb_xd, n, nbp));
    xdef = h->pb_xd;
  }
  if (h->rhsmap[level].set) {
    SWCHK(launch_ell(h, h->rhsmap[level], 0, xdef, nullptr, h->pb_rhs, nbp, T_OTHER));
    xdef = h->pb_rhs;
  }
  int total_f = 0, total_c = 0;
  SWCHK(solve_dev(h, fine_hid, level, xdef, h->pb_z, tol, maxiter, nbp, &total_f));
  SWCHK(stream_sync(h));
  if (level_is_direct(h, h->hier[fine_hid], level)) h->last_iters_f.assign(nb, 1);
  else SWCHK(record_iters(h, &h->hier[fine_hid].lv[level].sws, total_f, h->last_iters_f, nb));
  // xc = R x_def  (skip: R1 R0)             utils.py:298-304
  SWCHK(launch_ell(h, lv.R, 0, xdef, nullptr, h->pb_xc, nbp, T_R));
  const cplx* xc = h->pb_xc;
  if (skip) {
    SWCHK(launch_ell(h, H0.lv[1].R, 0, h->pb_xc, nullptr, h->pb_xc2, nbp, T_R));
    xc = h->pb_xc2;
  }
  // y = A_c^-1 xc                            utils.py:306-329
  SWCHK(solve_dev(h, 0, lcoarse, xc, h->pb_y, tol, maxiter, nbp, &total_c));
  SWCHK(stream_sync(h));
  if (lcoarse == H0.nlevels - 1 || level_is_direct(h, H0, lcoarse)) h->last_iters_c.assign(nb, 1);
  else SWCHK(record_iters(h, &H0.lv[lcoarse].sws, total_c, h->last_iters_c, nb));
  // w = P y (skip: P0 P1 y)                  utils.py:337-341
  const cplx* w;
  if (skip) {
    SWCHK(launch_ell(h, H0.lv[1].P, 0, h->pb_y, nullptr, h->pb_w2, nbp, T_P));
    SWCHK(launch_ell(h, lv.P, 0, h->pb_w2, nullptr, h->pb_w, nbp, T_P));
  } else {
    SWCHK(launch_ell(h, lv.P, 0, h->pb_y, nullptr, h->pb_w, nbp, T_P));
  }
  w = h->pb_w;
  // e = x0^H z - x0^H w                      utils.py:336,353-355
  SWCHK(dot_into(h, h->pb_x0, h->pb_z, n, nbp, h->pb_est + nbp));
  SWCHK(dot_into(h, h->pb_x0, w, n, nbp, h->pb_est + 2 * nbp));
  {
    LaunchScope ls(h, T_OTHER);
    hipLaunchKernelGGL(swk::k_est_combine, dim3((nbp + 255) / 256), dim3(256), 0, h->stream,
                       (const cplx*)(h->pb_est + nbp), (const cplx*)(h->pb_est + 2 * nbp), nbp,
                       h->pb_est);
    KLAUNCH_CHECK();
  }
  SWCHK(stream_sync(h));
  return 0;
}

// ---- the one collective of the path: trace-sum / variance statistics over the ranks (RCCL over xGMI)
namespace {
struct RcclUid {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*get_uid)(RcclUid*) = nullptr;
  int (*init_rank)(void**, int, RcclUid, int) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*destroy)(void*) = nullptr;
  bool tried = false, ok = false;
};
Rccl g_rccl;
bool load_rccl() {
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  // A process that runs torch.distributed with backend nccl already has RCCL mapped (PyTorch's bundled
  // copy); a second copy of a ROCm library in one process is what broke rocBLAS handle creation earlier
  // (two HIP runtimes' worth of static state), so: the mapped one first (RTLD_NOLOAD), by the names it
  // goes by, and only then a fresh load.
  for (const char* nm : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) g_rccl.lib = dlopen(nullptr, RTLD_NOW);
  if (!g_rccl.lib) g_rccl.lib = dlopen("librccl.so", RTLD_NOW);
  if (!g_rccl.lib) g_rccl.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW);
  if (!g_rccl.lib) return false;
  g_rccl.get_uid = (int (*)(RcclUid*))dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.init_rank = (int (*)(void**, int, RcclUid, int))dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(
      g_rccl.lib, "ncclAllReduce");
  g_rccl.destroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.ok = g_rccl.get_uid && g_rccl.init_rank && g_rccl.all_reduce && g_rccl.destroy;
  if (g_rccl.ok) g_rccl_destroy = g_rccl.destroy;
  return g_rccl.ok;
}
}  // namespace

int sw_comm_unique_id(char id[128]) {
  if (!id || !load_rccl()) return 1;
  RcclUid u;
  if (g_rccl.get_uid(&u) != 0) return 1;
  std::memcpy(id, u.internal, 128);
  return 0;
}

int sw_comm_init(sw_engine* h, int nranks, int rank, const char id[128]) {
  if (!h) return 1;
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return sw_fail(h, "bad arguments");
  if (!load_rccl()) return sw_fail(h, "RCCL could not be loaded (%s)", dlerror());
  HIPCHK(hipSetDevice(h->device));
  if (h->comm) {
    g_rccl.destroy(h->comm);
    h->comm = nullptr;
  }
  RcclUid u;
  std::memcpy(u.internal, id, 128);
  void* c = nullptr;
  const int rc = g_rccl.init_rank(&c, nranks, u, rank);
  if (rc != 0 || !c) return sw_fail(h, "ncclCommInitRank failed (status %d)", rc);
  h->comm = c;
  if (!h->d_stats) SWCHK(dev_realloc(h, &h->d_stats, (size_t)4));
  return 0;
}

int sw_allreduce_stats(sw_engine* h, double stats[4]) {
  if (!h) return 1;
  if (!stats) return sw_fail(h, "null statistics");
  if (!h->comm) return sw_fail(h, "no communicator (sw_comm_init)");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(h->d_stats, stats, 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int rc = g_rccl.all_reduce(h->d_stats, h->d_stats, 4, /*ncclFloat64*/ 8, /*ncclSum*/ 0,
                                   h->comm, h->stream);
  if (rc != 0) return sw_fail(h, "ncclAllReduce failed (status %d)", rc);
  HIPCHK(hipMemcpyAsync(stats, h->d_stats, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int sw_comm_destroy(sw_engine* h) {
  if (!h) return 1;
  if (h->comm && g_rccl.ok) g_rccl.destroy(h->comm);
  h->comm = nullptr;
  return 0;
}

int sw_sync(sw_engine* h) {
  if (!h) return 1;
  HIPCHK(hipSetDevice(h->device));
  if (h->gen_stream) HIPCHK(hipStreamSynchronize(h->gen_stream));
  return stream_sync(h);
}

int sw_hutch_fetch(sw_engine* h, double* ests, int32_t* iters) {
  if (!h) return 1;
  if (h->pb_nb <= 0 || !h->pb_est) return sw_fail(h, "nothing to fetch");
  HIPCHK(hipSetDevice(h->device));
  SWCHK(stream_sync(h));
  const int nb = h->pb_nb;
  if (ests) HIPCHK(hipMemcpy(ests, h->pb_est, sizeof(cplx) * nb, hipMemcpyDeviceToHost));
  if (iters) {
    for (int j = 0; j < nb; ++j) {
      iters[j] = j < (int)h->last_iters_f.size() ? h->last_iters_f[j] : 0;
      iters[nb + j] = j < (int)h->last_iters_c.size() ? h->last_iters_c[j] : 0;
    }
  }
  return 0;
}

int sw_hutch_batch(sw_engine* h, int mode, int level, int nb, const int8_t* probes, double tol,
                   int maxiter, double* ests, int32_t* iters) {
  SWCHK(sw_probes_upload(h, level, nb, probes));
  SWCHK(sw_hutch_run(h, mode, level, tol, maxiter));
  return sw_hutch_fetch(h, ests, iters);
}

// ---- measurement -----------------------------------------------------------------------
int sw_bench_dirac(sw_engine* h, int hid, int level, int nb, int reps, double* ms_per_apply) {
  SWCHK(check_hier(h, hid, level, true));
  if (nb <= 0 || reps <= 0 || !ms_per_apply) return sw_fail(h, "bad arguments");
  HIPCHK(hipSetDevice(h->device));
  Level& lv = h->hier[hid].lv[level];
  const int nbp = pad64(nb);
  cplx *a, *b;
  SWCHK(io_vectors(h, lv, nbp, &a, &b));
  // deterministic non-trivial input: +-1 pattern
  {
    std::vector<std::complex<double>> hx((size_t)lv.n * nbp);
    uint32_t s = 12345u;
    for (auto& v : hx) {
      s = s * 1664525u + 1013904223u;
      const double re = ((s >> 16) & 0xffff) / 65536.0 - 0.5;
      s = s * 1664525u + 1013904223u;
      const double im = ((s >> 16) & 0xffff) / 65536.0 - 0.5;
      v = std::complex<double>(re, im);
    }
    HIPCHK(hipMemcpy(a, hx.data(), hx.size() * sizeof(cplx), hipMemcpyHostToDevice));
  }
  const bool prof = h->profiling;
  h->profiling = false;
  const int bm = h->bench_mode;
  const int what = h->bench_what;
  Hier& H = h->hier[hid];
  if ((what == 1 || what == 2) && level + 1 >= H.nlevels) return sw_fail(h, "no transfer at this level");
  if (what == 1 || what == 2) SWCHK(ensure_level_ws(h, H.lv[level + 1], nbp));
  if (what == 3) SWCHK(ensure_level_ws(h, H.lv[H.nlevels - 1], nbp));
  cplx* c = lv.r;   // third buffer: the right-hand side of modes 1 / 2
  if (bm != 0) HIPCHK(hipMemcpy(c, a, (size_t)lv.n * nbp * sizeof(cplx), hipMemcpyDeviceToDevice));
  const cplx wgt = cplx{0.31, 0.07};
  auto one = [&](int i) -> int {
    if (what == 1) return launch_ell(h, lv.R, 0, a, nullptr, H.lv[level + 1].b, nbp, T_R);
    if (what == 2) return launch_ell(h, lv.P, 0, H.lv[level + 1].b, nullptr, b, nbp, T_P);
    if (what == 3) {
      Level& ll = H.lv[H.nlevels - 1];
      return apply_coarsest(h, H, (i & 1) ? ll.x : ll.b, (i & 1) ? ll.b : ll.x, nbp);
    }
    return apply_op(h, lv, bm, (i & 1) ? b : a, bm ? c : nullptr, (i & 1) ? a : b, nbp, wgt);
  };
  for (int i = 0; i < 3; ++i) SWCHK(one(i));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps; ++i) SWCHK(one(i));
  HIPCHK(hipEventRecord(e1, h->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  h->profiling = prof;
  *ms_per_apply = (double)ms / reps;
  return 0;
}

int sw_set_profiling(sw_engine* h, int on) {
  if (!h) return 1;
  if (h->gen_stream) HIPCHK(hipStreamSynchronize(h->gen_stream));
  SWCHK(stream_sync(h));
  h->profiling = on != 0;
  return 0;
}
int sw_timers(sw_engine* h, double t[8]) {
  if (!h || !t) return 1;
  SWCHK(stream_sync(h));
  for (int i = 0; i < 8; ++i) t[i] = h->tacc[i];
  t[T_MVM] += h->tacc[T_STENCIL] + h->tacc[T_STENCIL_RES] + h->tacc[T_STENCIL_SM] + h->tacc[T_MFMA_OP] +
              h->tacc[T_MFMA_OP2] + h->tacc[T_SCHUR] + h->tacc[T_SCHUR_OP];
  t[T_COARSEST] += h->tacc[T_MFMA_DENSE];
  return 0;
}
int sw_timers_reset(sw_engine* h) {
  if (!h) return 1;
  SWCHK(stream_sync(h));
  for (int i = 0; i < T_NCAT; ++i) {
    h->tacc[i] = 0.0;
    h->tcount[i] = 0;
    h->twork[i] = 0.0;
  }
  h->launches = 0;
  return 0;
}
int sw_kernel_stats(sw_engine* h, int which, double* total_ms, int64_t* launches) {
  if (!h || !total_ms || !launches) return 1;
  if (which < 0 || which >= T_NCAT) return sw_fail(h, "kernel class %d out of range", which);
  SWCHK(stream_sync(h));
  *total_ms = h->tacc[which];
  *launches = h->tcount[which];
  return 0;
}
int sw_kernel_work(sw_engine* h, int which, double* work) {
  if (!h || !work) return 1;
  if (which < 0 || which >= T_NCAT) return sw_fail(h, "kernel class %d out of range", which);
  *work = h->twork[which];
  return 0;
}
int sw_launch_count(sw_engine* h, int64_t* n) {
  if (!h || !n) return 1;
  *n = h->launches;
  return 0;
}

}  // extern "C"
