/*
 * schwinger_hip.h -- C ABI of the MI355X (gfx950) deflated-MLMC Hutchinson trace engine.
 *
 * This is the drop-in boundary for the hot path of Gustavroot/DeflatedMLMC_Schwinger
 * (SURVEY.md section 8).  The reference has no native code: every arithmetic call on the
 * path goes through SciPy/NumPy/pyamg wheels.  Each entry point below names the
 * reference call site(s) (file:line under the reference repo) whose work it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; sw_last_error() gives text;
 *     no C++ exception crosses the boundary;
 *   - complex numbers are interleaved (re, im) doubles ("complex128"); `double*` arguments
 *     that hold complex data say so;
 *   - host vectors use the REFERENCE ordering and shape: `nb` right-hand sides, each a
 *     contiguous flat vector of length n (level 0: idx(s,x,y) = s*L*L + y*L + x);
 *   - all buffers are caller-owned host memory unless the name ends in _dev;
 *   - one host thread per handle; device streams are internal;
 *   - a handle holds up to SW_MAX_HIER multigrid hierarchies (`hid`): hid 0 is the
 *     reference hierarchy (the MLMC level operators of multigrid.py:100-345), hid 1 an
 *     optional solver-only hierarchy for level 0 (parity is on converged solves).
 */
#ifndef SCHWINGER_HIP_H
#define SCHWINGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SW_MAX_HIER 2
#define SW_MAX_LEVELS 8
#define SW_MAX_KRYLOV 32
#define SW_MAX_DEFL 64

typedef struct sw_engine sw_engine;

/* ---- lifetime ------------------------------------------------------------------------ */
/* Replaces MG.__init__ (multigrid.py:58-89).  Fails (non-zero) when no HIP device exists. */
int sw_create(sw_engine** out, int device_id);
int sw_destroy(sw_engine* h);
const char* sw_last_error(sw_engine* h);          /* h may be NULL: last create() error */
int sw_device_count(void);                         /* 0 when no GPU is visible            */
/* Device blocks of 32 MB and more that an engine frees (setup scratch, workspaces, everything at sw_destroy)
 * are parked in a process-wide pool and handed out again instead of going back to the driver (cap SW_POOL_GB,
 * default 128 GB; 0 disables; an allocation that fails with blocks parked releases them and retries):
 * sw_pool_trim() releases what is parked. */
int sw_pool_trim(void);
const char* sw_version(void);

/* ---- operands (products of MG.setup, multigrid.py:100-345, and matrix.py:14-31) ------ */
/* Start (or reset) hierarchy `hid` with `nlevels` levels. */
int sw_hier_begin(sw_engine* h, int hid, int nlevels);
/* Level-0 operator as the matrix-free U(1) Wilson stencil A = S + mass*I of SURVEY F2
 * (replaces the CSR built by matrix.py:21-29).  U1,U2: complex128[L*L], site index y*L+x.
 * Sets n_0 = 2*L*L and the even-odd internal layout. */
int sw_set_lattice(sw_engine* h, int hid, int L, double mass, const double* U1, const double* U2);
/* Level operator as CSR, complex128 data (multigrid.py:276-280 A_{l+1} = R A P; also
 * accepted for level 0 when no lattice is set). */
int sw_set_csr(sw_engine* h, int hid, int level, int n, const int64_t* indptr,
               const int32_t* indices, const double* data);
/* Prolongator P_l (n_f x n_c CSR, multigrid.py:262-264); R_l = P_l^H (multigrid.py:267-274). */
int sw_set_transfer(sw_engine* h, int hid, int level, int n_f, int n_c, const int64_t* indptr,
                    const int32_t* indices, const double* data);
/* Dense inverse of the coarsest operator, row-major complex128[n*n] (multigrid.py:342-344). */
int sw_set_coarsest_inv(sw_engine* h, int hid, int n, const double* dense);
/* Cycle shape at `level`: MR smoothing steps before/after the coarse correction and the
 * number of flexible-GMRES steps wrapped around the next-level cycle (0 = plain V-cycle).
 * Stands in for lgmres(maxiter=smooth_iters) at multigrid.py:393-394,438-439 (parity is on
 * converged solves). */
int sw_set_cycle(sw_engine* h, int hid, int level, int nu_pre, int nu_post, int kcycle);
/* Replace the adaptive MR steps at `level` by a fixed polynomial smoother: n_pre / n_post
 * Richardson steps x <- x + w_k (b - A x) with complex128 weights w_k (the inverse roots of a
 * GMRES residual polynomial computed at setup).  One fused kernel per step, no inner products.
 * n_pre = n_post = 0 returns to MR(nu) of sw_set_cycle. */
int sw_set_smoother(sw_engine* h, int hid, int level, int n_pre, const double* w_pre, int n_post,
                    const double* w_post);
/* Even-odd (Schur-complement) post-smoother for the stencil level: n_post Richardson steps
 * x_e <- x_e + w_k (b'_e - S x_e) on the even sites, S = D - H_eo H_oe / D, then the odd sites
 * exactly (x_o = (b_o + H_oe x_e) / D).  Half-vector traffic per step, operator four times better
 * conditioned; takes the place of the post-smoothing steps of sw_set_smoother on that level (no
 * pre-smoothing).  n_post = 0 switches back.  Same role as sw_set_smoother
 * (multigrid.py:438-439). */
int sw_set_eo_smoother(sw_engine* h, int hid, int level, int n_post, const double* w_post);
/* The same construction on a coarse (block) level, whose operator is a 5-point stencil of 16x16
 * blocks over the coarse sites: four operators that act on the even or odd sites only, in MFMA
 * block-row form -- which = 0: S = D_ee - A_eo D_oo^-1 A_oe (9-point), 1: F = A_eo D_oo^-1,
 * 2: G = D_oo^-1, 3: Hb = D_oo^-1 A_oe.  RT row tiles, KS k-steps per tile (multiple of 4),
 * tmap[RT] = tile (site) of the level vector each row tile writes, kcol[RT*KS] = first level row of
 * each 4-column group, vals[RT*KS*64] complex128 lane-packed as for the level operators.  With all
 * four set, sw_set_eo_smoother accepts the level.  Optional which = 4: the DENSE inverse of S over the even
 * sites' rows (every row tile lists all even sites' columns); with it the level is solved exactly --
 * x_e = S^-1 (b_e - F b_o), x_o = G b_o - Hb x_e -- instead of smoothed and the levels below it are not
 * visited (option "eo_direct"); it is dropped whenever one of the four operators is set again.
 * sw_get_level_bsr reads a (device-built) level operator back in that form (kcol = vals = NULL: only *KS). */
int sw_set_eo_operator(sw_engine* h, int hid, int level, int which, int RT, int KS, const int32_t* tmap,
                       const int32_t* kcol, const double* vals);
int sw_get_level_bsr(sw_engine* h, int hid, int level, int* KS, int32_t* kcol, double* vals);
/* The same four operators built ON THE DEVICE from the level's own block-row operator (as
 * sw_setup_galerkin leaves it: five site blocks per row, own site last; Lc x Lc sites, Lc even and >= 8):
 * batched 16x16 inverses of the odd sites' diagonal blocks and block products, no host algebra.
 * sw_apply_eo_operator applies one of them (which as above) to nb full-length level vectors in the
 * reference layout (rows it does not write come back zero) -- used to fit the smoother polynomial to S
 * and by the parity tests.  Build-only extension of the even-odd smoother above. */
int sw_setup_eo_operators(sw_engine* h, int hid, int level, int Lc);
int sw_apply_eo_operator(sw_engine* h, int hid, int level, int which, int nb, const double* X, double* Y);
/* Reference-faithful cycle at `level` ("function_iters / total_complexity in both
 * modes"): MG.one_mg_step exactly as multigrid.py:369-447 lays it out -- smooth, residual,
 * restrict, recurse, prolong, residual, smooth -- with `cycles` restart cycles of unpreconditioned
 * GMRES(m) from a zero guess as the smoother, the counterpart of
 * lgmres(A_l, r, tol=1e-20, maxiter=smooth_iters=2) with SciPy's inner_m = 30 (SURVEY F5;
 * LGMRES's single augmentation vector in the second cycle is not reproduced).  m = 0 switches back. */
int sw_set_gmres_smoother(sw_engine* h, int hid, int level, int m, int cycles);
/* ---- GPU-side setup of a hierarchy (device counterpart of multigrid.py:157-280) ----
 * The test vectors of a level stay in HBM.  Coarse levels built this way carry 16 dofs per coarse
 * site (8 test vectors x 2 chirality halves), i.e. one MFMA row tile per site.
 * sw_setup_testvectors: `sweeps` rounds of V <- A_level^-1 V by batched flexible GMRES to `tol`
 *   (precond = 0: unpreconditioned, for a hierarchy still under construction; 1: with the finished
 *   hierarchy's own cycle from `level` down).  seed != 0 starts from pseudo-random vectors, seed = 0
 *   from the vectors the level already holds (the restricted ones of the level above).
 *   iters_out[sweeps] receives the iteration counts.  Replaces eigs(A_l, k, sigma=0), multigrid.py:174.
 * sw_setup_transfer: per-(aggregate, half) orthonormalisation of the level's test vectors on the
 *   device (multigrid.py:232-259), P_level and R_level = P^H from it (multigrid.py:262-274), and
 *   the restricted test vectors as level+1's start.  blk_rows[nblocks*rpb]: level rows (engine row
 *   order) of each block; P's grouped-ELL structure comes from the caller's geometry: group size
 *   G, K columns per group pcols[ngroups*K], and pmap[ngroups*K*G] = index (block*rpb + member)*8 + k
 *   of the value, or -1; porder[ngroups] (may be NULL): order in which the kernel visits the row
 *   groups (groups that share coarse columns adjacent).
 * sw_setup_galerkin: A_{level+1} = R A P (multigrid.py:276-280) by 16-colour probing with the
 *   engine's own operator kernels, written straight into MFMA block-row form.  nbr[ncs*5]: the five
 *   sites of each coarse site's 5-point neighbourhood, pairwise distinct (the site itself last lets
 *   the smoother kernel keep its own X rows in registers).
 * sw_get_level_dense: a device-built level operator as a dense row-major complex128[n*n] (for the
 *   host inverse of the coarsest level, multigrid.py:342-344). */
int sw_setup_testvectors(sw_engine* h, int hid, int level, int nvec, uint64_t seed, int sweeps,
                         double tol, int maxiter, int precond, int32_t* iters_out);
int sw_setup_transfer(sw_engine* h, int hid, int level, int nblocks, int rpb, const int32_t* blk_rows,
                      int G, int K, const int32_t* pcols, const int64_t* pmap, const int32_t* porder);
int sw_setup_galerkin(sw_engine* h, int hid, int level, int Lc, const int32_t* nbr);
int sw_get_level_dense(sw_engine* h, int hid, int level, double* dense);
/* Dense inverse of a device-built coarsest operator without leaving the GPU: in-place Gauss-Jordan
 * with partial pivoting by the engine's own kernels (k_gj_*; n <= 8192), then packed into MFMA block-row
 * form as sw_set_coarsest_inv would (multigrid.py:342-344: np.linalg.inv).  Fails on a singular operator. */
int sw_setup_invert_coarsest(sw_engine* h, int hid);
/* The dense coarsest inverse the engine holds (handed over by sw_set_coarsest_inv or formed on the device by
 * sw_setup_invert_coarsest -- which also accepts a coarsest operator given as CSR, sw_set_csr) as a row-major
 * complex128[n*n] host array: the reference's mg_solver.coarsest_inv (multigrid.py:342-344) for callers that
 * read it (stoch_trace.py:428-435 traces it directly). */
int sw_get_coarsest_inv(sw_engine* h, int hid, double* dense);
/* The same one level up: the DENSE inverse of a block level's even-odd Schur complement (operator 0 of
 * sw_setup_eo_operators / sw_set_eo_operator; at most 8192 rows) formed on the device and installed as
 * that level's even-odd operator 4, with which the cycle solves the level exactly (option "eo_direct")
 * instead of smoothing it and visiting the levels below (multigrid.py:342-344,413-416 one level up). */
int sw_setup_direct_level(sw_engine* h, int hid, int level);
/* Dense inverse of a small level's OPERATOR (n <= 8192, multiple of 16), formed on the device and kept as the
 * level's direct solver (option "direct_small", default on): solves that START at this level -- the MLMC
 * coarse solves A_c^-1 R x of utils.py:306-329 on the reference hierarchy's small levels, the fine solves of
 * its coarse difference levels (utils.py:292) -- become x = A^-1 b, x += A^-1 (b - A x) on the matrix cores
 * instead of a multigrid-preconditioned FGMRES (multigrid.py:347-366); same solution to the solver tolerance,
 * iteration count reported as 1. */
int sw_setup_level_inverse(sw_engine* h, int hid, int level);
/* Arnoldi relation for the smoother polynomial, on the device: `degree` steps (classical Gram-Schmidt
 * twice) of the level operator (which = 0) or of its even-odd Schur complement (which = 1) from a
 * pseudo-random start vector; Hout receives the (degree+1) x degree Hessenberg matrix, row-major
 * complex128.  The host turns its harmonic Ritz values into the weights of sw_set_smoother /
 * sw_set_eo_smoother (the tuned stand-in for lgmres(maxiter=2), multigrid.py:393-394). */
int sw_setup_arnoldi(sw_engine* h, int hid, int level, int which, int degree, uint64_t seed, double* Hout);
/* ---- device eigensolver (SURVEY 8f-2 completed: the host ARPACK + SuperLU calls of the setup) -------------
 * Block subspace iteration with the engine's own batched solve as the shift-invert operator: replaces
 * eigs(A_l, k, sigma=0) (multigrid.py:174: test vectors) and eigsh(gamma_3 A, k, sigma=0) (utils.py:140:
 * deflation vectors).  The engine does the O(n) work on blocks of 64 vectors held in three device buffers
 * (index 0..2) on one finished (hid, level); the caller does the 64 x 64 dense algebra in between
 * (Rayleigh-Ritz on V^H Op^-1 V, Cholesky-QR), see setup_gpu.device_eigenpairs.
 *   sw_eig_begin   buffers on (hid, level), buffer 0 <- pseudo-random block
 *   sw_eig_load    first ncols columns of buffer dst <- host vectors (reference order)
 *   sw_eig_solve   buf_dst = Op^-1 buf_src, 64 right-hand sides to `tol`; mode 0: Op = A_level, mode 1:
 *                  Op = gamma_3 A_level (gamma_3 = +1 / -1 on the first / second half of the reference order)
 *   sw_eig_gram    out[64*64] = buf_a^H buf_b (fp64 MFMA, deterministic two-stage sum)
 *   sw_eig_rotate  buf_dst = buf_src Y (sub < 0) or buf_sub - buf_src Y (the block residual W - V T), Y[64*64]
 *                  row-major, dst != src
 *   sw_eig_fetch   first k columns of buf_src as k host vectors in the reference order
 *   sw_eig_end     release the buffers */
int sw_eig_begin(sw_engine* h, int hid, int level, uint64_t seed);
int sw_eig_load(sw_engine* h, int dst, int ncols, const double* X);
int sw_eig_solve(sw_engine* h, int src, int dst, int mode, double tol, int maxiter, int32_t* iters_max);
int sw_eig_gram(sw_engine* h, int a, int b, double* out);
int sw_eig_rotate(sw_engine* h, int src, const double* Y, int dst, int sub);
int sw_eig_fetch(sw_engine* h, int src, int k, double* out);
int sw_eig_end(sw_engine* h);
/* Mark the hierarchy complete (allocates level workspaces lazily). */
int sw_hier_end(sw_engine* h, int hid);

/* Deflation vectors U (utils.py:145-155), row-major complex128[n0*k], reference ordering. */
int sw_set_deflation(sw_engine* h, int k, const double* U);
/* MLMC-level deflation vectors V_l of the difference operator at `level` of hid 0
 * (utils.py:141-157,260-266), row-major complex128[n_l*k]; k = 0 clears. */
int sw_set_level_deflation(sw_engine* h, int level, int k, const double* V);
/* Index shift of Pperm at `level` of hid 0 (multigrid.py:142-155,320-326). shift<0 clears. */
int sw_set_perm(sw_engine* h, int level, int64_t shift);
/* MLMC right-hand-side map C_i = Bblock_perm_i * Pperm_i^T as CSR (multigrid.py:328-331,
 * utils.py:288-290); n x n at `level` of hid 0. */
int sw_set_rhsmap(sw_engine* h, int level, int n, const int64_t* indptr, const int32_t* indices,
                  const double* data);
/* Outer flexible-GMRES restart length (<= SW_MAX_KRYLOV) and the hierarchy used to
 * precondition level-0 solves (0 or 1). */
int sw_set_solver(sw_engine* h, int restart, int solver_hid);

/* Engine switches (defaults are the measured-best settings; everything else exists for A/B runs,
 * profiles/ holds the measurements).  Unknown names and out-of-range values fail with a message.
 *   solver:        "eo_direct" (1) block levels that were given the dense inverse of their even-odd Schur
 *                  complement (sw_set_eo_operator, which = 4) are solved with it instead of smoothed;
 *                  "eo_solve" (1) outer solves of an even-odd smoothed lattice level on the even-odd
 *                  reduced system (half-length Krylov vectors; same stopping criterion, same results);
 *                  "precond_f32" (0) multigrid cycle in complex64 inside the fp64 FGMRES; "f32_krylov" (1)
 *                  with it, complex64 Krylov basis per restart cycle; "cgs2" (0) / "inner_cgs2" (0) second
 *                  Gram-Schmidt pass; "pyth_last" (1) last Arnoldi step of a restart cycle without its
 *                  orthogonalisation pass; "verify" (1) true-residual check of every outer solve;
 *                  "stop_factor" (1) outer solves iterate until every residual is below stop_factor * tol
 *                  (iteration counts are still reported at tol; 0.1 pins per-probe estimates to 1e-10
 *                  relative even where they cancel to small numbers); "fused_reduce" (1) inner products
 *                  completed inside the launch that forms their partial sums, FGMRES scalar updates
 *                  riding along; "gram_cycle" (1) restart cycles of the even-odd reduced outer solve and the
 *                  K-cycle's inner iteration in Gram-matrix form (directions without orthogonalisation, one
 *                  pass for all inner products, per-probe Cholesky solve); "lgmres_aug" (1) LGMRES
 *                  augmentation vector in the reference-faithful smoother's second cycle;
 *                  "lazy_sync" (1) convergence read-back only near the expected iteration count;
 *                  "dot_blocks" row blocks of the reducing BLAS-1 launches
 *   stencil level: "stencil_spw", "stencil_tile", "stencil_nt" (sites per wave, lattice tile width,
 *                  non-temporal stores); "p_even" (1) prolongation onto the even sites only ahead of an
 *                  even-odd smoother; "eo_skew" (-1) time-skewed strip order of the even-odd smoother's steps
 *                  on lattices beyond the Infinity Cache (-1 automatic, 0 off, > 0 strip height in rows; batches
 *                  too wide for admissible strips are walked 64-probe chunk by chunk, "eo_skew_chunk" (0)
 *                  forces that);
 *                  "eo_tile" (0) the fp64 even-odd launches of the lattice level that carry no b' operand (reduced
                  operator, product-form factors) from LDS-staged halo tiles in rotated coordinates, double-buffered
                  by LDS-DMA, one persistent workgroup of 4 or 8 waves per CU (k_schur_tile); bit-identical results;
                  "eo_product" (1) the even-odd smoother of the reduced-system cycle in product form,
 *                  x + beta prod_j (1 - u_j S)(b' - S x): the same polynomial as the steps
 *                  x <- x + w_k (b' - S x), 2 nu + 2 half-vector passes instead of 3 nu
 *   block levels:  "use_mfma" (1) fp64-MFMA block-row kernels vs grouped ELL; "mfma_3m" (1) three real
 *                  matrix products per complex one (k_bsr_mfma3) instead of four; "mfma3_tiles" (0: by size)
 *                  tiles of 16 probes per wave in that kernel; "mfma_ops", "mfma_tiles",
 *                  "mfma_small_tiles", "bsr_stages" / "dense_stages" (register pipeline depth), "bsr_nt",
 *                  "bsr_xreg", "bsr_sub", "dense_map", "ell_order"; complex64 twins "f32_tiles",
 *                  "f32_stages", "f32_dense_stages", "f32_splitk", "f32_pairs"
 *                  "dense_lds" (4) dense operators (coarsest inverse, dense Schur inverse of a directly solved
 *                  level, direct inverse of a small level) through k_dense_mfma3_lds: operands shared through LDS
 *                  (register-staged, double-buffered), 2 or 4 row tiles x 2 probe tiles per workgroup (0: k_bsr_mfma3)
 *   setup:         "gj_block" (32) panel width of the blocked Gauss-Jordan inverse (0: unblocked)
 *   sw_bench_dirac: "bench_mode" (0 Y=AX, 1 residual, 2 smoother step), "bench_what" (operator / R / P /
 *                  coarsest) */
int sw_set_option(sw_engine* h, const char* name, double value);
/* Current value of a switch (save / restore around A/B runs); also the read-only counter
 * "direct_fallbacks": solves on a directly solved level (sw_setup_level_inverse) whose measured residual stayed
 * above the tolerance after four refinement steps and that were handed to the iterative path instead. */
int sw_get_option(sw_engine* h, const char* name, double* value);

/* ---- building blocks (host buffers, reference ordering) -------------------------------- */
/* Y = A_level X.  Replaces MG.matvec (multigrid.py:552-557) and the residual SpMVs at
 * multigrid.py:388,402,433.  X,Y: complex128[nb*n]. */
int sw_apply_dirac(sw_engine* h, int hid, int level, int nb, const double* X, double* Y);
/* Y = R_level X (multigrid.py:406; utils.py:301-303) / Y = P_level X (multigrid.py:429;
 * utils.py:339-341). */
int sw_restrict(sw_engine* h, int hid, int level, int nb, const double* X, double* Y);
int sw_prolong(sw_engine* h, int hid, int level, int nb, const double* X, double* Y);
/* Y = coarsest_inv X (multigrid.py:413-416; utils.py:309-310,321-322). */
int sw_coarsest(sw_engine* h, int hid, int nb, const double* X, double* Y);
/* X = one multigrid cycle applied to B starting at level0 (MG.one_mg_step, multigrid.py:369-447). */
int sw_vcycle(sw_engine* h, int hid, int level0, int nb, const double* B, double* X);
/* Solve A_level0 X = B to ||r|| < tol*||b|| per right-hand side (MG.solve -> pyamg fgmres,
 * multigrid.py:347-366).  iters[nb], relres[nb] may be NULL.  Non-convergence within
 * maxiter is NOT an error (the reference discards exitCode); it is visible in relres. */
int sw_solve(sw_engine* h, int hid, int level0, int nb, const double* B, double* X, double tol,
             int maxiter, int32_t* iters, double* relres);

/* ---- the probe loop body ----------------------------------------------------------------- */
#define SW_MODE_HUTCHINSON 0   /* utils.py:210-250 */
#define SW_MODE_MLMC 1         /* utils.py:252-361 */
#define SW_MODE_MLMC_SKIP 2    /* utils.py:252-361 with mg_solver.skip_level and i == 0 */
#define SW_MODE_LEVEL 3        /* e_k = x^H A_l^-1 C_l x: one level's own trace term, estimated
                                  stochastically (build-only; the reference computes the coarsest
                                  term directly and raises for the stochastic form,
                                  stoch_trace.py:428-437) */
/* One batch of probes x_k in {-1,+1}^n (int8, nb*n, reference ordering) at `level`
 * (build-only extension: entries +-2 encode +-i, i.e. Z4 probes {1,i,-1,-i}):
 *   HUTCHINSON: e_k = x^H A^-1 Pperm^T (x - U U^H x)
 *   MLMC:       e_k = x^H A_f^-1 C x - x^H P A_c^-1 R C x   (SKIP: P0 P1, R1 R0)
 * ests: complex128[nb]; iters: int32[2*nb] = fine-solve and coarse-solve iteration counts. */
int sw_hutch_batch(sw_engine* h, int mode, int level, int nb, const int8_t* probes, double tol,
                   int maxiter, double* ests, int32_t* iters);
/* Split form used when the probes are to be resident in HBM before timing starts:
 * upload -> run (asynchronous on the engine stream) -> sync -> fetch. */
int sw_probes_upload(sw_engine* h, int level, int nb, const int8_t* probes);   /* slot 0 + select */
/* Several batches resident at once (bench: all inputs in HBM before the timed region). */
int sw_probes_upload_slot(sw_engine* h, int slot, int level, int nb, const int8_t* probes);
int sw_probes_select(sw_engine* h, int slot);
/* Device-side probe generation (replaces np.random.randint(2, size=n) at utils.py:213-216,
 * 255-258 for probes that never need to exist on the host).  The engine holds a window of the
 * reference's MT19937 stream: sw_probes_stream_set() hands over the 624 raw state words at the
 * stream position that becomes position 0 (sw_mt_window() below makes them from a seed or from a
 * NumPy state); sw_probes_generate() fills `slot` with the nb probes whose first entry is draw
 * number `pos` of that stream (one 32-bit draw per entry, bit-exact with NumPy), jumping there
 * with GF(2) jump polynomials -- cost independent of `pos`.  Asynchronous on a generation stream of the
 * engine's own: the call returns when the work is queued, the sw_hutch_run that consumes the slot waits
 * for it on the device -- so the probes of batch k + 1 (another slot) are drawn while batch k is solved. */
#define SW_PROBES_Z2 1   /* entry = 2*(draw & 1) - 1              (the reference's probes)          */
#define SW_PROBES_Z4 2   /* draw & 3 -> 1, i, -1, -i as 1,2,-1,-2  (build-only, BASELINE config 1)  */
int sw_probes_stream_set(sw_engine* h, const uint32_t* window);
int sw_probes_generate(sw_engine* h, int slot, int level, int nb, int kind, uint64_t pos);
/* Copy the int8 codes held in `slot` (uploaded or generated) to the host: nb*n bytes. */
int sw_probes_fetch(sw_engine* h, int slot, int8_t* out);
int sw_hutch_run(sw_engine* h, int mode, int level, double tol, int maxiter);
int sw_sync(sw_engine* h);
int sw_hutch_fetch(sw_engine* h, double* ests, int32_t* iters);

/* ---- multi-GPU: the one collective of the path ------------------------------------ */
/* One process per GPU, one engine per process; the probe loop shards by probe and needs a single
 * small reduction per round: {sum Re e, sum Im e, sum |e|^2, n} (population variance as
 * stoch_trace.py:143-145).  RCCL (ncclAllReduce over xGMI) behind the C ABI, loaded on first use:
 * rank 0 calls sw_comm_unique_id and hands the 128 bytes to the other ranks by any means; every
 * rank calls sw_comm_init; sw_allreduce_stats sums the four doubles in place over all ranks. */
int sw_comm_unique_id(char id[128]);
int sw_comm_init(sw_engine* h, int nranks, int rank, const char id[128]);
int sw_allreduce_stats(sw_engine* h, double stats[4]);
int sw_comm_destroy(sw_engine* h);

/* ---- measurement ----------------------------------------------------------------------- */
/* Timed stencil loop for the roofline figure: `reps` applications of the level-0 operator of
 * hierarchy hid on nb resident right-hand sides, HIP events on the engine stream.
 * ms_per_apply is the average launch duration. */
int sw_bench_dirac(sw_engine* h, int hid, int level, int nb, int reps, double* ms_per_apply);
/* Bucketed device time (ms) since the last reset, measured with HIP events when profiling
 * is enabled (CustomTimer buckets of utils.py:366-445 plus the Krylov BLAS-1 the reference
 * leaves untimed):  [0]=mvm [1]=defl [2]=P [3]=R [4]=axpy [5]=dots [6]=coarsest [7]=other. */
int sw_set_profiling(sw_engine* h, int on);
int sw_timers(sw_engine* h, double t[8]);
int sw_timers_reset(sw_engine* h);
/* Accumulated HIP-event time and launch count of one kernel class since the last reset
 * (profiling on): classes 0..7 as sw_timers (without the kernels listed next), and */
#define SW_KCLASS_STENCIL 8         /* k_stencil<0>  Y = A X                       */
#define SW_KCLASS_STENCIL_RES 9     /* k_stencil<1>  Y = B - A X                   */
#define SW_KCLASS_STENCIL_SM 10     /* k_stencil<2>  Y = X + w (B - A X)           */
#define SW_KCLASS_MFMA_DENSE 11     /* k_bsr_mfma, dense coarsest inverse          */
#define SW_KCLASS_MFMA_OP 12        /* k_bsr_mfma, block-structured level operator */
/* (class 13: unused; it was a fused two-step stencil, removed after it measured no faster) */
#define SW_KCLASS_MFMA_OP2 14       /* k_bsr_mfma, level operators below level 1   */
#define SW_KCLASS_SCHUR 15          /* k_schur_step / k_eo_hop, even-odd smoother  */
int sw_kernel_stats(sw_engine* h, int which, double* total_ms, int64_t* launches);
/* Floating-point operations issued by the launches of an MFMA kernel class since the last reset
 * (profiling on): 8 flops per complex multiply-add over every (row tile, k-step, probe). */
int sw_kernel_work(sw_engine* h, int which, double* work);
/* Kernel launches issued since the last reset (for launch-bound analysis). */
int sw_launch_count(sw_engine* h, int64_t* n);

/* ---- host-only helpers (no GPU needed) ---------------------------------------------------- */
/* The probe stream of utils.py:213-216: MT19937 seeded as np.random.seed(seed); entry =
 * 2*(next_uint32 & 1) - 1; the stream continues across calls. */
typedef struct sw_mt19937 sw_mt19937;
sw_mt19937* sw_mt_create(uint32_t seed);
void sw_mt_destroy(sw_mt19937* g);
void sw_mt_skip(sw_mt19937* g, uint64_t ndraws);
void sw_mt_raw(sw_mt19937* g, uint64_t n, uint32_t* out);
void sw_mt_rademacher(sw_mt19937* g, uint64_t n, int8_t* out);
void sw_mt_z4(sw_mt19937* g, uint64_t n, int8_t* out);   /* codes 1,2,-1,-2 from draw & 3 */
/* NumPy interchange: key[624] + pos exactly as np.random.get_state()[1:3]. */
sw_mt19937* sw_mt_from_state(const uint32_t* key, int pos);
void sw_mt_get_state(const sw_mt19937* g, uint32_t* key, int* pos);
/* The 624 raw words starting at the current stream position (input of sw_probes_stream_set). */
void sw_mt_window(sw_mt19937* g, uint32_t* out);
/* Jump ahead by ndraws in O(1) state refills: g_J(x) = x^J mod phi(x) over GF(2) applied to the
 * window (sw_mt_skip is the sequential checker).  sw_mt_jump_poly returns the 19937 coefficient
 * bits of g_J in 624 words (bit i of the array = coefficient of x^i); non-zero = failure. */
int sw_mt_jump(sw_mt19937* g, uint64_t ndraws);
int sw_mt_jump_poly(uint64_t ndraws, uint32_t* poly);
int sw_mt_window_jump(const uint32_t* window, uint64_t ndraws, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* SCHWINGER_HIP_H */
