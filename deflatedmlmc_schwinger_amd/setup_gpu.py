"""GPU-side setup of the level-0 preconditioner hierarchy.

The reference obtains its test vectors from ARPACK with a SuperLU shift-invert
(multigrid.py:174).  That is kept for the reference hierarchy (it defines the MLMC levels), but
for the solver-only hierarchy the same quality of coarse space comes from a few sweeps of
*inverse iteration on random vectors*, v <- A_l^-1 v, done as batched (inexact) solves on the
engine: no sparse factorisation, setup cost of a few batched solves, and it scales to lattices
where SuperLU on the host is impractical.  Galerkin products and the per-aggregate QR stay on
the host (SciPy / NumPy).
"""
import sys
import os
import time

import numpy as np
import scipy.sparse as sp

from . import engine as _engine
from . import hierarchy as _hier


def _inverse_iterate(eng, hid, level0, V, sweeps, tol, maxiter):
    """`sweeps` rounds of V <- orth(A^-1 V) with inexact batched solves on the engine."""
    its_log = []
    for _ in range(sweeps):
        X, its, _ = eng.solve(hid, level0, np.ascontiguousarray(V.T), tol, maxiter)
        X = np.atleast_2d(X)
        its_log.append(int(np.max(its)))
        V, _ = np.linalg.qr(X.T)
    return V, its_log


def adaptive_solver_hierarchy(eng, A0, lat, cfg, hid):
    """Build {A, P, coarsest_inv} for `cfg["coarsening"]` with engine-side inverse iteration.
    Uses hierarchy slot `hid` of `eng` as scratch (it is redefined by the caller afterwards)."""
    L, mass, U1, U2 = lat
    sweeps = int(cfg.get("setup_sweeps", 3))
    tol = float(cfg.get("setup_tol", 1.0e-2))
    maxiter = int(cfg.get("setup_maxiter", 400))
    rng = np.random.default_rng(int(cfg.get("setup_seed", 7)))
    As = [sp.csr_matrix(A0).astype(np.complex128)]
    Ps, tvs, log = [], [], []
    Lf, hd = L, 1
    V = None
    t0 = time.time()
    for lvl, (agg, nvec) in enumerate(cfg["coarsening"]):
        n = As[-1].shape[0]
        if Lf % agg:
            raise Exception("lattice extent %d not divisible by aggregate edge %d" % (Lf, agg))
        if V is None:
            V = rng.standard_normal((n, nvec)) + 1j * rng.standard_normal((n, nvec))
        elif V.shape[1] < nvec:
            extra = nvec - V.shape[1]
            V = np.hstack([V, rng.standard_normal((n, extra)) + 1j * rng.standard_normal((n, extra))])
        V = np.ascontiguousarray(V[:, :nvec])
        # the level operator alone in the scratch slot: unpreconditioned batched GMRES(32)
        eng.hier_begin(hid, 1)
        if lvl == 0:
            eng.set_lattice(hid, L, mass, U1, U2)
        else:
            eng.set_csr(hid, 0, As[-1])
        eng.hier_end(hid)
        eng.set_solver(32, hid)
        V, its = _inverse_iterate(eng, hid, 0, V, sweeps, tol, maxiter)
        log.append({"level": lvl, "n": n, "gmres_iterations": its})
        tvs.append(V)
        P = _hier._site_prolongator(As[-1], Lf, hd, agg, nvec, V, lvl == 0)
        Ac = sp.csr_matrix(P.conjugate().transpose() @ As[-1] @ P)
        Ps.append(P)
        As.append(Ac)
        V = np.asarray(P.conjugate().transpose() @ V)       # coarse image of the test vectors
        Lf //= agg
        hd = nvec
    cinv = np.linalg.inv(As[-1].toarray())
    return {"A": As, "P": Ps, "coarsest_inv": cinv, "tv": tvs, "cfg": cfg,
            "setup_log": log, "setup_s": time.time() - t0}


# ----------------------------------------------------------------------------------------------
# everything on the device: test vectors, per-aggregate QR, P / R, Galerkin products
# ----------------------------------------------------------------------------------------------
def level_geometry(Lf, hd, agg, fine_level):
    """Index geometry of one coarsening step (agg x agg site aggregates with a chirality split, as
    hierarchy._site_prolongator): which engine rows form each (aggregate, half) block, and the
    grouped-ELL structure of the prolongator over them.

    Level 0 rows are in the engine's even-odd order ((par*V/2 + site/2)*2 + half), coarse levels in
    site-major order ((site*2 + half)*hd + k).  Returns dict(blk_rows[nblocks, rpb], G, pcols[ng, K],
    pmap[ng, K, G], Lc, nbr[Lc*Lc, 5])."""
    if Lf % agg:
        raise Exception("lattice extent %d not divisible by aggregate edge %d" % (Lf, agg))
    V = Lf * Lf
    n = 2 * V * hd
    idx = np.arange(n, dtype=np.int64)
    if fine_level:
        if hd != 1:
            raise Exception("level 0 carries one dof per spin")
        half = idx // V
        site = idx % V
        x, y = site % Lf, site // Lf
        internal = (((x + y) & 1) * (V // 2) + (site >> 1)) * 2 + half
    else:
        site = idx // (2 * hd)
        half = (idx // hd) % 2
        x, y = site % Lf, site // Lf
        internal = idx
    Lc = Lf // agg
    block = ((y // agg) * Lc + (x // agg)) * 2 + half
    nblocks = 2 * Lc * Lc
    rpb = n // nblocks
    order = np.argsort(block, kind="stable")
    blk_rows = internal[order].reshape(nblocks, rpb)
    blk_of = np.empty(n, dtype=np.int64)
    pos_of = np.empty(n, dtype=np.int64)
    blk_of[blk_rows] = np.arange(nblocks)[:, None]
    pos_of[blk_rows] = np.arange(rpb)[None, :]
    # rows per group of the prolongator.  Level 0: 8 consecutive rows = four sites of one parity along x, both
    # spins, where they stay inside one aggregate (aggregate edges that are multiples of 8): the group's 16 coarse
    # rows are then pulled through the L2 once per 8 fine rows instead of once per 4 (k_ell<8,0> 30 us against
    # k_ell<4,0> 35 us per launch at 128^2 x 256 probes, profiles/r04_ab_sessions.txt r04aa); else 4 (two sites)
    for G in ((int(os.environ.get("SW_P_GROUP0", "8")), 4) if fine_level else (8,)):
        ng = n // G
        rb = blk_of.reshape(ng, G)
        b0, b1 = rb.min(axis=1), rb.max(axis=1)
        if ((rb == b0[:, None]) | (rb == b1[:, None])).all():
            break
    else:
        raise Exception("a row group of the prolongator touches more than two blocks")
    two = bool((b0 != b1).any())
    K = 16 if two else 8
    k8 = np.arange(8)
    pcols = np.empty((ng, K), dtype=np.int64)
    pcols[:, :8] = b0[:, None] * 8 + k8[None, :]
    if two:
        pcols[:, 8:] = b1[:, None] * 8 + k8[None, :]
    rows = np.arange(n).reshape(ng, G)                       # engine row of (group, g)
    src = (blk_of[rows] * rpb + pos_of[rows]) * 8            # [ng, G]
    pmap = np.full((ng, K, G), -1, dtype=np.int64)
    own0 = rb == b0[:, None]
    pmap[:, :8, :] = np.where(own0[:, None, :], src[:, None, :] + k8[None, :, None], -1)
    if two:
        own1 = (rb == b1[:, None]) & (b1 != b0)[:, None]
        pmap[:, 8:, :] = np.where(own1[:, None, :], src[:, None, :] + k8[None, :, None], -1)
    cs = np.arange(Lc * Lc)
    xc, yc = cs % Lc, cs // Lc
    nbr = np.stack([cs, yc * Lc + (xc + 1) % Lc, yc * Lc + (xc - 1) % Lc,
                    ((yc + 1) % Lc) * Lc + xc, ((yc - 1) % Lc) * Lc + xc], axis=1)
    nbr = np.sort(nbr, axis=1)
    if Lc > 1:
        # ... the site itself LAST: the smoother kernel then finds its own X rows in the operand
        # registers of the final four k-steps (k_bsr_mfma, xreg)
        nbr = np.concatenate([nbr[nbr != cs[:, None]].reshape(-1, 4), cs[:, None]], axis=1)
    # visit the row groups sorted by their first coarse column: the groups of one aggregate (spread
    # over both parity halves of the even-odd order at level 0) become neighbours in time and L2
    porder = np.argsort(pcols[:, 0], kind="stable")
    if (porder == np.arange(ng)).all():
        porder = None
    return {"blk_rows": blk_rows, "G": G, "pcols": pcols, "pmap": pmap, "porder": porder, "Lc": Lc,
            "nbr": nbr, "n_c": nblocks * 8}


class _EngineOperator:
    """A level operator of an engine hierarchy as something hierarchy.smoother_weights can use."""

    def __init__(self, eng, hid, level, n):
        self.eng, self.hid, self.level = eng, hid, level
        self.shape = (n, n)

    def __matmul__(self, v):
        return self.eng.apply_dirac(self.hid, self.level, np.asarray(v, dtype=np.complex128))


class _EngineSchur:
    """Even-odd Schur complement S = D - A_eo A_oe / D of the level-0 operator of an engine
    hierarchy, matrix-free through sw_apply_dirac (for hierarchy.smoother_weights)."""

    def __init__(self, eng, hid, L, mass):
        V = L * L
        idx = np.arange(2 * V)
        site = idx % V
        even = (((site % L) + (site // L)) & 1) == 0
        self.E, self.O = idx[even], idx[~even]
        self.D = 4.0 + mass
        self.eng, self.hid, self.n = eng, hid, 2 * V
        self.shape = (self.E.size, self.E.size)

    def __matmul__(self, v):
        u = np.zeros(self.n, dtype=np.complex128)
        u[self.E] = v
        w = self.eng.apply_dirac(self.hid, 0, u)
        u[:] = 0.0
        u[self.O] = w[self.O] / self.D
        w = self.eng.apply_dirac(self.hid, 0, u)
        return self.D * np.asarray(v) - w[self.E]


class _EngineBlockSchur:
    """S = D_ee - A_eo D_oo^-1 A_oe of a block level as the engine holds it (sw_setup_eo_operators),
    acting on the even sites' rows, through sw_apply_eo_operator (for hierarchy.smoother_weights)."""

    def __init__(self, eng, hid, level, Lc):
        site = np.arange(Lc * Lc)
        even = (((site % Lc) + (site // Lc)) & 1) == 0
        self.E = np.nonzero(np.repeat(even, 16))[0]
        self.eng, self.hid, self.level, self.n = eng, hid, level, Lc * Lc * 16
        self.shape = (self.E.size, self.E.size)

    def __matmul__(self, v):
        u = np.zeros(self.n, dtype=np.complex128)
        u[self.E] = v
        return self.eng.apply_eo_operator(self.hid, self.level, 0, u)[self.E]


def device_solver_hierarchy(eng, lat, cfg, hid):
    """The solver hierarchy built ON THE GPU (solver_cfg["setup"] = "device"): test vectors by
    batched inverse iteration (first unpreconditioned on each new level, then one refinement pass
    preconditioned by the hierarchy itself), per-aggregate QR, P / R and the Galerkin operators
    A_{l+1} = R A P by colour probing, all with engine kernels; the host only supplies index
    geometry, inverts the coarsest operator and finds the smoother polynomials.
    Requires 8 test vectors per half on every coarsening and coarse extents that are multiples of 4."""
    L, mass, U1, U2 = lat
    coarsening = [tuple(c) for c in cfg["coarsening"]]
    for agg, nvec in coarsening:
        if nvec != 8:
            raise Exception("the device setup needs 8 test vectors per chirality half")
    nl = len(coarsening) + 1
    sweeps = int(cfg.get("setup_sweeps", 3))
    tol = float(cfg.get("setup_tol", 0.1))
    maxiter = int(cfg.get("setup_maxiter", 32))
    refine = int(cfg.get("setup_refine", 1))
    seed = int(cfg.get("setup_seed", 7))
    eo_levels = _hier.eo_levels_of(cfg)
    t0 = time.time()
    log = []
    geo = []
    clock = {"geometry": 0.0, "testvectors": 0.0, "transfer+galerkin": 0.0, "coarsest_inverse": 0.0,
             "smoother_polynomials": 0.0}

    # every phase: wall seconds and, of those, the host seconds inside hipMalloc / hipFree (the engine counts
    # them: option "alloc_seconds"); the difference is kernels + host round trips
    alloc_clock = {}

    class _timed:
        def __init__(self, key):
            self.key = key

        def __enter__(self):
            self.t = time.time()
            self.a = eng.get_option("alloc_seconds")

        def __exit__(self, *exc):
            clock[self.key] += time.time() - self.t
            alloc_clock[self.key] = alloc_clock.get(self.key, 0.0) + eng.get_option("alloc_seconds") - self.a

    Lf, hd = L, 1
    with _timed("geometry"):
        for lvl, (agg, nvec) in enumerate(coarsening):
            geo.append(level_geometry(Lf, hd, agg, lvl == 0))
            Lf //= agg
            hd = nvec
    sizes = [2 * L * L] + [g["n_c"] for g in geo]

    def rebuild_operators(level):
        """P_l, R_l, A_{l+1} for l >= level from the test vectors in place (device only)."""
        with _timed("transfer+galerkin"):
            for lv in range(level, nl - 1):
                g = geo[lv]
                eng.setup_transfer(hid, lv, g["blk_rows"], g["G"], g["pcols"], g["pmap"], g["porder"])
                eng.setup_galerkin(hid, lv, g["Lc"], g["nbr"])

    def finish():
        """coarsest inverse (host, as multigrid.py:342-344), cycle shapes, smoother polynomials."""
        with _timed("coarsest_inverse"):
            done = False
            if cfg.get("setup_inverse", "device") == "device":
                try:
                    eng.setup_invert_coarsest(hid)             # Gauss-Jordan on the GPU (k_gj_*)
                    done = True
                except _engine.EngineError as err:
                    # refused (size, singular pivot): LAPACK on the host decides (setup only, never
                    # the solve path)
                    sys.stderr.write("setup_gpu: %s -- inverting the coarsest operator on the host\n" % err)
            if not done:
                eng.set_coarsest_inv(hid, _hier.dense_inverse(eng.level_dense(hid, nl - 1)))
            eng.hier_end(hid)
        with _timed("smoother_polynomials"):
            on_device = (cfg.get("setup_polynomials", "device") == "device"
                         and cfg.get("smoother_target", "all") == "all")

            def fit(lv, which, degree, host_op):
                """weights of `degree` steps for the level operator (which 0) / its Schur complement (1)"""
                if degree <= 0:
                    return np.zeros(0, dtype=np.complex128)
                if on_device and degree <= 32:
                    # Arnoldi on the GPU (sw_setup_arnoldi, at most 32 vectors): the host sees a
                    # (degree+1) x degree matrix; longer polynomials are fitted through the C ABI below
                    return _hier.weights_from_hessenberg(eng.setup_arnoldi(hid, lv, which, degree))
                proj = None
                if which == 0 and cfg.get("smoother_target", "all") == "complement":
                    # polynomial fitted on what the coarse correction leaves: v - P R v
                    proj = (lambda v, _lv=lv: v - eng.prolong(hid, _lv, eng.restrict(hid, _lv, v)))
                return _hier.smoother_weights(host_op(), degree, project=proj)

            for lv in range(nl - 1):
                cyc = cfg["cycle"][lv]
                eng.set_cycle(hid, lv, cyc[0], cyc[1], cyc[2])
                if cfg.get("smoother", "richardson") == "richardson":
                    eo_here = lv in eo_levels
                    full_op = (lambda _lv=lv: _EngineOperator(eng, hid, _lv, sizes[_lv]))
                    # (an even-odd smoothed level without pre-smoothing never runs its full-operator
                    # post-smoother: vcycle_rich takes the even-odd one; with pre-smoothing it does run)
                    eng.set_smoother(hid, lv, fit(lv, 0, cyc[0], full_op),
                                     fit(lv, 0, 0 if eo_here and on_device and cyc[0] == 0 else cyc[1], full_op))
                    if lv == 0 and eo_here:
                        eng.set_eo_smoother(hid, 0, fit(0, 1, cyc[1],
                                                        lambda: _EngineSchur(eng, hid, L, mass)))
                    elif eo_here:
                        Lc_l = geo[lv - 1]["Lc"]
                        if Lc_l >= 8 and cfg.get("setup_eo", "device") == "device":
                            # block level: S, F, G, Hb by batched 16 x 16 algebra on the device, the
                            # polynomial fitted to S through the engine
                            eng.setup_eo_operators(hid, lv, Lc_l)
                            eng.set_eo_smoother(hid, lv, fit(
                                lv, 1, cyc[1], lambda _lv=lv, _Lc=Lc_l: _EngineBlockSchur(eng, hid, _lv, _Lc)))
                        else:
                            # tiny lattices: the operator comes back in block-row form, the four
                            # operators are formed on the host and uploaded
                            _hier.upload_coarse_eo([eng], hid, lv, eng.level_bsr(hid, lv), Lc_l, cyc[1])

    eng.hier_begin(hid, nl)
    eng.set_lattice(hid, L, mass, U1, U2)
    eng.set_solver(int(cfg.get("restart", 24)), hid)      # fixes the Krylov workspace shape
    # pass 1: top-down, each new level's test vectors by unpreconditioned inverse iteration
    for lvl in range(nl - 1):
        t1 = time.time()
        with _timed("testvectors"):
            its = eng.setup_testvectors(hid, lvl, 8, seed if lvl == 0 else 0, sweeps, tol, maxiter,
                                        False)
        # tol = 0: the sweeps are RELAXATION (a fixed number of unpreconditioned GMRES(m) steps that damp
        # the rough components of the random vectors), not solves; the converged inverse-iteration
        # solves are those of pass 2, preconditioned by the hierarchy itself
        log.append({"pass": 1, "level": lvl, "n": sizes[lvl],
                    ("relaxation_steps" if tol <= 0.0 else "gmres_iterations"): its,
                    "seconds": round(time.time() - t1, 3)})
        g = geo[lvl]
        with _timed("transfer+galerkin"):
            eng.setup_transfer(hid, lvl, g["blk_rows"], g["G"], g["pcols"], g["pmap"], g["porder"])
            eng.setup_galerkin(hid, lvl, g["Lc"], g["nbr"])
    finish()
    eng.set_solver(int(cfg.get("restart", 24)), hid)
    # pass 2: refinement, preconditioned by the hierarchy itself (a few iterations per solve)
    refine_levels = int(cfg.get("setup_refine_levels", 1))     # how many levels (from the top) refine
    for _ in range(refine):
        for lvl in range(min(refine_levels, nl - 1)):
            t1 = time.time()
            with _timed("testvectors"):
                its = eng.setup_testvectors(hid, lvl, 8, 0, 1, float(cfg.get("setup_refine_tol", 1e-2)),
                                            int(cfg.get("setup_refine_maxiter", 64)), True)
            log.append({"pass": 2, "level": lvl, "n": sizes[lvl], "gmres_iterations": its,
                        "seconds": round(time.time() - t1, 3)})
            # new P_lvl => new coarse basis: every operator below is rebuilt from the restricted
            # vectors, and the dense coarsest inverse with them
            rebuild_operators(lvl)
            finish()
    # optional: block levels solved exactly in even-odd reduced form (cfg["direct_levels"]): the dense
    # inverse of the level's Schur complement becomes its even-odd operator 4; the Schur steps of that
    # level and everything below it drop out of the cycle.  Formed on the device (sw_setup_direct_level);
    # cfg["setup_direct"] = "host" keeps the LAPACK route for comparison.
    clock["direct_inverse"] = 0.0
    with _timed("direct_inverse"):
        for lv in cfg.get("direct_levels", ()):
            lv = int(lv)
            if not (1 <= lv < nl - 1) or lv not in eo_levels:
                raise Exception("direct_levels: level %d is not an even-odd smoothed block level" % lv)
            Lc_l = geo[lv - 1]["Lc"]
            if cfg.get("setup_direct", "device") == "device":
                eng.setup_direct_level(hid, lv)          # S -> dense -> Gauss-Jordan, on the GPU
                continue
            sb = _hier.site_blocks_from_block_rows(*eng.level_bsr(hid, lv))
            ops = _hier.coarse_schur_blocks(sb[0], sb[1], Lc_l) if sb is not None else None
            if ops is None:
                raise Exception("direct_levels: level %d has no 5-point block structure" % lv)
            eng.set_eo_operator(hid, lv, 4, *_hier.dense_schur_inverse_blocks(ops))
    log.append({"seconds": {k: round(v, 3) for k, v in clock.items()},
                "of_which_hipMalloc_hipFree": {k: round(v, 3) for k, v in alloc_clock.items()},
                "allocated_GB_so_far": round(eng.get_option("alloc_gbytes"), 2)})
    return {"levels": sizes, "setup_log": log, "setup_s": time.time() - t0, "cfg": cfg}


# ----------------------------------------------------------------------------------------------
# device eigensolver: the host ARPACK + SuperLU calls of the setup, on the GPU
# ----------------------------------------------------------------------------------------------
def device_eigenpairs(eng, hid, level, k, tol, hermitian_g3=False, maxit=60, seed=11, start=None,
                      log=None):
    """The k eigenpairs nearest zero of A_level (hermitian_g3 = False: what eigs(A_l, k, sigma=0, tol)
    returns at multigrid.py:174) or of Q = gamma_3 A_level (True: eigsh(Q, k, sigma=0, tol), utils.py:140),
    by block subspace iteration with Rayleigh-Ritz on the engine:

        V orthonormal [n][64];  W = Op^-1 V (the engine's batched multigrid solve: the shift-invert
        operator ARPACK gets from SuperLU);  T = V^H W (fp64 MFMA Gram kernel);  Ritz pairs (theta, y) of T;
        block residual E = W - V T on the device, M = E^H E: |Op^-1 x - theta x|^2 = y^H M y for x = V y
        (formed from the small residual vectors themselves -- the difference W^H W - T^H T of two O(theta^2)
        matrices would lose everything below 1e-8);  V <- W R^-1 (Cholesky-QR, twice).

    Stops when the k Ritz pairs of largest |theta| all satisfy ARPACK's own shift-invert criterion
    |Op^-1 x - theta x| <= tol |theta|.  The block width 64 (the engine's batch quantum: 64 solves cost what
    one costs) makes the convergence factor |lambda_k / lambda_65| per step (0.13 for A, 0.16 for Q on
    schwinger128).  The solves start loose and tighten with the residual (inexact inverse iteration).
    Returns (lambda[k], X[n, k]) -- X columns of unit norm, orthonormal for the hermitian case --, and
    appends per-step records to `log`."""
    import scipy.linalg as sla
    m = 64
    if not 1 <= k <= m // 2:
        raise Exception("device_eigenpairs: k = %d outside 1..%d" % (k, m // 2))
    mode = 1 if hermitian_g3 else 0
    eng.eig_begin(hid, level, seed)
    try:
        cur, nxt, tmp = 0, 1, 2
        if start is not None:
            eng.eig_load(cur, np.asarray(start))

        def cholqr(src, dst):
            """dst = src R^-1 with R^H R = src^H src"""
            G = eng.eig_gram(src, src)
            R = sla.cholesky(0.5 * (G + G.conj().T), lower=False)
            eng.eig_rotate(src, sla.solve_triangular(R, np.eye(m, dtype=np.complex128), lower=False), dst)

        # orthonormal start block
        cholqr(cur, nxt)
        cholqr(nxt, cur)
        res_prev = 1.0
        theta = Y = None
        for it in range(maxit):
            # the shift-invert solve; its tolerance follows the residual of the wanted pairs (inexact inverse
            # iteration: an error of tol_s in b - A x moves the wanted Ritz vectors by about
            # tol_s |lambda_k / lambda_1|, so two orders below the current residual is ample)
            tol_s = min(1e-3, max(5e-13, 1e-2 * res_prev))
            its = eng.eig_solve(cur, nxt, mode, tol_s)
            T = eng.eig_gram(cur, nxt)
            eng.eig_rotate(cur, T, tmp, sub=nxt)            # E = W - V T
            M = eng.eig_gram(tmp, tmp)
            if hermitian_g3:
                theta, Y = np.linalg.eigh(0.5 * (T + T.conj().T))
            else:
                theta, Y = np.linalg.eig(T)
                Y = Y / np.linalg.norm(Y, axis=0)[None, :]
            # nearest zero = largest |theta|; of an (almost) degenerate modulus -- a complex-conjugate pair
            # of A -- the member with the positive imaginary part first (ARPACK's choice is implementation-
            # defined there; this one is deterministic)
            lam_all = 1.0 / theta
            order = np.lexsort((-np.sign(np.round(lam_all.imag, 12)), np.round(np.abs(lam_all), 10)))
            theta, Y = theta[order], Y[:, order]
            yk = Y[:, :k]
            r2 = np.einsum("ik,ij,jk->k", yk.conj(), M, yk).real
            if not hermitian_g3:
                # T y = theta y holds only to the accuracy of the dense eigensolver: add what it leaves
                r2 = r2 + np.linalg.norm(T @ yk - yk * theta[None, :k], axis=0) ** 2
            res = np.sqrt(np.maximum(r2, 0.0)) / np.abs(theta[:k])
            res_prev = float(res.max())
            if log is not None:
                log.append({"step": it, "solve_tol": tol_s, "solve_iterations": its, "residual_max": res_prev})
            if res_prev <= tol:
                break
            cholqr(nxt, tmp)
            cholqr(tmp, cur)
        else:
            raise Exception("device_eigenpairs: %d wanted pairs not converged to %g in %d steps (residual %.2e)"
                            % (k, tol, maxit, res_prev))
        Yfull = np.zeros((m, m), dtype=np.complex128)
        Yfull[:, :k] = Y[:, :k]
        eng.eig_rotate(cur, Yfull, tmp)
        X = eng.eig_fetch(tmp, k).T
        lam = 1.0 / theta[:k]
        if hermitian_g3:
            lam = lam.real
        return lam, np.ascontiguousarray(X)
    finally:
        eng.eig_end()
