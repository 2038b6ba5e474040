"""GPU-side setup of the level-0 preconditioner hierarchy.

The reference obtains its test vectors from ARPACK with a SuperLU shift-invert
(multigrid.py:174).  That is kept for the reference hierarchy (it defines the MLMC levels), but
for the solver-only hierarchy the same quality of coarse space comes from a few sweeps of
*inverse iteration on random vectors*, v <- A_l^-1 v, done as batched (inexact) solves on the
engine: no sparse factorisation, setup cost of a few batched solves, and it scales to lattices
where SuperLU on the host is impractical.  Galerkin products and the per-aggregate QR stay on
the host (SciPy / NumPy).
"""
import time

import numpy as np
import scipy.sparse as sp

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
