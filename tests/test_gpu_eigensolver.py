"""GPU tests of the device eigensolver (sw_eig_* + setup_gpu.device_eigenpairs): the replacement of the
setup's host ARPACK + SuperLU calls -- eigs(A_l, k, sigma=0) at multigrid.py:174 (test vectors) and
eigsh(gamma_3 A, k, sigma=0) at utils.py:140 (deflation vectors) -- against SciPy's ARPACK on the same
operators.  Tolerances: eigenvalues 1e-8 relative (both sides are iterated to 1e-9), residuals
|A x - lambda x| <= 1e-8 |x| (ARPACK's shift-invert criterion at tol 1e-9 bounds them by tol * |A|),
eigenspaces to 1e-7 in the sine of the largest principal angle; block kernels against NumPy at 1e-13."""
import time

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu

from deflatedmlmc_schwinger_amd import gateway, hierarchy, matrix, setup_gpu, utils  # noqa: E402
from deflatedmlmc_schwinger_amd.multigrid import MG, REF_HID, SOLVER_HID  # noqa: E402


def _solver_only(name="schwinger128"):
    params = gateway.set_params(name)
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    lat = hierarchy.detect_lattice(A)
    mg = MG(A)
    mg.setup_solver_only(hierarchy.auto_solver_cfg(lat[0]), device=0, engines=1)
    return A, mg


def test_block_kernels_against_numpy():
    """k_block_gram (fp64 MFMA), k_block_rotate (also in residual form) and the gamma_3 solve mode, through
    the C ABI, on the 128^2 lattice level."""
    A, mg = _solver_only()
    eng = mg.engine
    n = A.shape[0]
    rng = np.random.default_rng(3)
    V = rng.standard_normal((64, n)) + 1j * rng.standard_normal((64, n))
    W = rng.standard_normal((64, n)) + 1j * rng.standard_normal((64, n))
    eng.eig_begin(SOLVER_HID, 0)
    try:
        eng.eig_load(0, V)
        eng.eig_load(1, W)
        G = eng.eig_gram(0, 1)
        ref = V.conj() @ W.T
        assert np.abs(G - ref).max() / np.abs(ref).max() < 1e-13
        Y = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
        eng.eig_rotate(0, Y, 2)
        out = eng.eig_fetch(2, 64)
        ref = (V.T @ Y).T
        assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-13
        eng.eig_rotate(0, Y, 2, sub=1)
        out = eng.eig_fetch(2, 64)
        assert np.abs(out - (W - ref)).max() / np.abs(ref).max() < 1e-13
        # buf2 = (gamma_3 A)^-1 buf0  <=>  gamma_3 A buf2 = buf0
        eng.eig_solve(0, 2, 1, 1e-12)
        X = eng.eig_fetch(2, 8)
        g3 = np.ones(n)
        g3[n // 2:] = -1.0
        back = (A @ X.T) * g3[:, None]
        assert np.linalg.norm(back - V[:8].T) / np.linalg.norm(V[:8]) < 1e-11
    finally:
        eng.eig_end()
    eng.close()


def test_device_eigenpairs_match_arpack_on_schwinger128():
    A, mg = _solver_only()
    eng = mg.engine
    n = A.shape[0]
    # ---- eigs(A, k = 4, sigma = 0, tol = 1e-9): the level-0 test vectors of the reference hierarchy
    log = []
    t0 = time.time()
    lam, X = setup_gpu.device_eigenpairs(eng, SOLVER_HID, 0, 4, 1e-9, log=log)
    t_dev = time.time() - t0
    t0 = time.time()
    w, v = spla.eigs(sp.csc_matrix(A), k=4, which="LM", tol=1e-9, maxiter=1000000, sigma=0.0)
    t_host = time.time() - t0
    print("eigs k=4: device %.3f s (%d steps, %d solver iterations), host ARPACK %.3f s"
          % (t_dev, len(log), sum(r["solve_iterations"] for r in log), t_host))
    print(log)
    # the fourth eigenvalue is one member of a complex-conjugate pair: compare moduli and
    # real parts, and |imaginary part|
    ow, ol = np.argsort(np.abs(w)), np.argsort(np.abs(lam))
    w, v, lam, X = w[ow], v[:, ow], lam[ol], X[:, ol]
    assert np.max(np.abs(np.abs(w) - np.abs(lam)) / np.abs(w)) < 1e-8
    assert np.max(np.abs(w.real - lam.real) / np.abs(w)) < 1e-8
    assert np.max(np.abs(np.abs(w.imag) - np.abs(lam.imag)) / np.abs(w)) < 1e-8
    assert lam[3].imag > 0.0                                   # the deterministic choice of the pair
    assert np.max(np.abs(np.linalg.norm(X, axis=0) - 1.0)) < 1e-12
    res = np.linalg.norm(A @ X - X * lam[None, :], axis=0)
    print("eigs residuals |A x - lambda x|:", res)
    assert res.max() < 1e-8
    # the three real eigenvalues' vectors agree with ARPACK's up to a phase
    for j in range(3):
        c = abs(np.vdot(v[:, j], X[:, j]))
        assert abs(c - 1.0) < 1e-12, (j, c)
    # ---- eigsh(gamma_3 A, k = 8, sigma = 0, tol = 1e-9): the deflation vectors
    g3 = np.ones(n)
    g3[n // 2:] = -1.0
    Q = (sp.diags(g3) @ A).tocsc()
    log = []
    t0 = time.time()
    lamq, Xq = setup_gpu.device_eigenpairs(eng, SOLVER_HID, 0, 8, 1e-9, hermitian_g3=True, log=log)
    t_dev = time.time() - t0
    t0 = time.time()
    S, Vq = spla.eigsh(Q, k=8, which='LM', tol=1e-9, sigma=0.0)
    t_host = time.time() - t0
    print("eigsh k=8: device %.3f s (%d steps, %d solver iterations), host ARPACK %.3f s"
          % (t_dev, len(log), sum(r["solve_iterations"] for r in log), t_host))
    print(log)
    assert np.max(np.abs(np.sort(S) - np.sort(lamq)) / np.abs(np.sort(S))) < 1e-9
    assert np.abs(Xq.conj().T @ Xq - np.eye(8)).max() < 1e-12
    res = np.linalg.norm(Q @ Xq - Xq * lamq[None, :], axis=0)
    print("eigsh residuals |Q x - lambda x|:", res)
    assert res.max() < 1e-8
    # same invariant subspace: sine of the largest principal angle
    sv = np.linalg.svd(Vq.conj().T @ Xq, compute_uv=False)
    assert np.sqrt(max(0.0, 1.0 - sv.min() ** 2)) < 1e-7
    # the low-rank trace term of utils.py:145-173 from the device vectors against the golden tr1
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
    tr1_gold = complex(*gold["defl128_tr1"])
    sgn = np.where(lamq > 0, 1.0, -1.0)
    shift = 128 * 2 * 2
    Ux = (Xq * sgn[None, :]) * g3[:, None]
    Ux = np.roll(Ux, -shift, axis=0)           # Pperm * U: (Pperm v)[r] = v[(r + shift) mod n]
    tr1 = np.sum(np.einsum("ik,ik->k", Ux.conj(), Xq) / np.abs(lamq))
    print("tr1 device %r golden %r" % (tr1, tr1_gold))
    assert abs(tr1 - tr1_gold) / abs(tr1_gold) < 1e-8
    eng.close()


def test_hutchinson_flow_with_deferred_coarse_levels_gives_the_same_result():
    """stoch_trace.hutchinson builds the coarse levels of the reference hierarchy on a host thread WHILE the
    probes run (MG.setup with defer_coarse_levels, joined by MG.finish_setup for the work model): same trace,
    stopping index, iteration total and complexity figure as with the levels built up front; and the device
    setup (eigensolves on the GPU) against the reference's host ARPACK setup: same stopping index, traces equal
    to 1e-7 (two eigensolvers at 1e-9 behind the deflation vectors)."""
    import contextlib
    import io
    from deflatedmlmc_schwinger_amd import stoch_trace
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    out = {}
    for key, extra in (("deferred", {}), ("upfront", {"defer_coarse_levels": False}),
                       ("reference", {"setup_eigs": "reference"})):
        tp = utils.trace_params_from_params(dict(params, **extra), "hutchinson")
        tp.update(extra)
        with contextlib.redirect_stdout(io.StringIO()):
            out[key] = stoch_trace.hutchinson(A, tp)
    a, b, c = out["deferred"], out["upfront"], out["reference"]
    assert a['nr_ests'] == b['nr_ests'] == c['nr_ests']
    assert a['function_iters'] == b['function_iters']
    assert abs(a['trace'] - b['trace']) <= 1e-12 * abs(b['trace'])
    assert a['total_complexity'] == b['total_complexity']
    assert abs(a['trace'] - c['trace']) <= 1e-7 * abs(c['trace'])
    assert abs(a['total_complexity'] - c['total_complexity']) <= 1e-3 * c['total_complexity']
