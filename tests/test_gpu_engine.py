"""GPU parity tests (run with ``-m gpu`` on an MI355X): every C-ABI entry point of the hot
path against the CPU oracle on the same seeded inputs.

Tolerances: the path is complex128 floating point.  Single operator applications are compared
at 1e-13 relative (different summation order than CSR), per-probe estimates at the north-star
tolerance of 1e-10 relative against the sparse-LU oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from deflatedmlmc_schwinger_amd import gateway, matrix, utils  # noqa: E402
from deflatedmlmc_schwinger_amd.engine import (MODE_HUTCHINSON, MODE_LEVEL, MODE_MLMC,  # noqa: E402
                                               MODE_MLMC_SKIP)
from deflatedmlmc_schwinger_amd.multigrid import MG, REF_HID, SOLVER_HID  # noqa: E402
from oracle import engine_model as em  # noqa: E402
from oracle import ref_path as rp  # noqa: E402


def _rand(shape, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def _relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


class Problem:
    def __init__(self, name, k_defl, batch_solver_cfg=None):
        params = gateway.set_params(name)
        params['function_tol'] = 1e-12
        self.params = params
        self.A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
        self.tp = utils.trace_params_from_params(params, "mlmc")
        self.tp['nr_deflat_vctrs'] = k_defl
        self.tp['mlmc_deflat_vctrs'] = [0] * 3
        # the building-block tests below compare against the host-built default hierarchy; the
        # drop-in flows (gateway presets, no solver_cfg) get hierarchy.auto_solver_cfg
        from deflatedmlmc_schwinger_amd import hierarchy as _h
        self.tp['solver_cfg'] = dict(_h.DEFAULT_SOLVER_CFG) if batch_solver_cfg is None else batch_solver_cfg
        self.mg = MG(self.A)
        self.mg.setup(dof=self.tp['dof'], aggrs=self.tp['aggrs'],
                      max_levels=self.tp['max_nr_levels'], dim=2,
                      acc_eigvs=self.tp['accuracy_mg_eigvs'], sys_type='schwinger', params=self.tp)
        self.mg.total_levels = len(self.mg.ml.levels)
        self.Ux, self.tr1 = utils.deflation_pre_computations(
            self.A, k_defl, 1e-9, "hutchinson", self.mg.timer, self.tp, self.mg)
        self.levels = self.mg.ml.levels
        self.eng = self.mg.engine
        self.lu = {}

    def lu_solver(self, level):
        if level not in self.lu:
            self.lu[level] = rp.LUSolver(self.levels[level].A)
        return self.lu[level]


@pytest.fixture(scope="module")
def p16():
    return Problem('schwinger16', 8)


@pytest.fixture(scope="module")
def p128():
    return Problem('schwinger128', 8)


@pytest.mark.parametrize("nb", [1, 3, 70])
def test_dirac_stencil_matches_csr_16(p16, nb):
    X = _rand((nb, p16.A.shape[0]), 1)
    Y = p16.eng.apply_dirac(REF_HID, 0, X)
    ref = (p16.A @ X.T).T
    assert _relerr(Y, ref) < 1e-13


def test_dirac_stencil_matches_csr_128(p128):
    X = _rand((5, p128.A.shape[0]), 2)
    Y = p128.eng.apply_dirac(REF_HID, 0, X)
    ref = (p128.A @ X.T).T
    assert _relerr(Y, ref) < 1e-13
    # same operator registered in the solver hierarchy
    Y2 = p128.eng.apply_dirac(SOLVER_HID, 0, X)
    assert np.array_equal(Y, Y2)


def test_dirac_linearity_full_batch(p128):
    """size-independent property at the benchmark batch size (nb = 256): A(aX+Y) = aAX+AY."""
    n = p128.A.shape[0]
    X, Y = _rand((256, n), 3), _rand((256, n), 4)
    a = 0.7 - 0.3j
    lhs = p128.eng.apply_dirac(REF_HID, 0, a * X + Y)
    rhs = a * p128.eng.apply_dirac(REF_HID, 0, X) + p128.eng.apply_dirac(REF_HID, 0, Y)
    assert _relerr(lhs, rhs) < 1e-13
    # and one column against the CSR matrix
    assert _relerr(lhs[17], p128.A @ (a * X[17] + Y[17])) < 1e-13


@pytest.mark.parametrize("prob", ["p16", "p128"])
def test_coarse_operators_and_transfers(prob, request):
    p = request.getfixturevalue(prob)
    nlev = len(p.levels)
    for l in range(nlev):
        n = p.levels[l].A.shape[0]
        X = _rand((3, n), 10 + l)
        Y = p.eng.apply_dirac(REF_HID, l, X)
        assert _relerr(Y, (p.levels[l].A @ X.T).T) < 1e-13, "A level %d" % l
        if l < nlev - 1:
            Yr = p.eng.restrict(REF_HID, l, X)
            assert _relerr(Yr, (p.levels[l].R @ X.T).T) < 1e-13, "R level %d" % l
            Xc = _rand((3, p.levels[l + 1].A.shape[0]), 20 + l)
            Yp = p.eng.prolong(REF_HID, l, Xc)
            assert _relerr(Yp, (p.levels[l].P @ Xc.T).T) < 1e-13, "P level %d" % l
    Xc = _rand((4, p.levels[-1].A.shape[0]), 30)
    Yc = p.eng.coarsest(REF_HID, Xc)
    assert _relerr(Yc, (np.asarray(p.mg.coarsest_inv) @ Xc.T).T) < 1e-12


def test_vcycle_matches_numpy_model(p16, p128):
    for p in (p16, p128):
        As = [l.A for l in p.levels]
        Ps = [l.P for l in p.levels[:-1]]
        post = 4
        cfg = [(0, post, 0)] * (len(As) - 1)
        for level0 in range(len(As) - 1):
            B = _rand((As[level0].shape[0], 3), 40 + level0)
            ref = em.cycle(As, Ps, p.mg.coarsest_inv, cfg, level0, B)
            X = p.eng.vcycle(REF_HID, level0, B.T.copy())
            assert _relerr(X.T, ref) < 1e-10, "cycle from level %d" % level0


def test_solver_cycle_polynomial_smoother_matches_numpy_model(p16, p128):
    """the level-0 preconditioner cycle (fixed-polynomial smoother fused into the operator
    kernels) against the NumPy model with the same weights."""
    for p in (p16, p128):
        sh = p.mg.solver_hier
        cfg = [tuple(c) for c in sh["cfg"]["cycle"]]
        B = _rand((sh["A"][0].shape[0], 3), 45)
        ref = em.cycle(sh["A"], sh["P"], sh["coarsest_inv"], cfg, 0, B, weights=p.mg.solver_weights)
        X = p.eng.vcycle(SOLVER_HID, 0, B.T.copy())
        assert _relerr(X.T, ref) < 1e-10


def test_solve_reaches_tolerance_and_matches_lu(p16, p128):
    for p, tol_x in ((p16, 1e-9), (p128, 1e-8)):
        n = p.A.shape[0]
        B = _rand((6, n), 50)
        X, iters, relres = p.mg.solve_batch(0, B, 1e-12)
        true_rel = np.linalg.norm(B.T - p.A @ X.T, axis=0) / np.linalg.norm(B.T, axis=0)
        assert true_rel.max() < 5e-12, true_rel
        assert relres.max() < 1e-12
        assert iters.min() >= 1
        ref = np.stack([p.lu_solver(0)(B[k]) for k in range(B.shape[0])])
        assert _relerr(X, ref) < tol_x
    # solves that start on a coarse level of the reference hierarchy (MLMC level-2 solves)
    n2 = p128.levels[2].A.shape[0]
    B = _rand((4, n2), 51)
    X, iters, relres = p128.mg.solve_batch(2, B, 1e-12)
    true_rel = np.linalg.norm(B.T - p128.levels[2].A @ X.T, axis=0) / np.linalg.norm(B.T, axis=0)
    assert true_rel.max() < 5e-12


def test_cycle_end_step_without_orthogonalisation_pass(p16, p128):
    """option pyth_last (default on): the last Arnoldi step of every restart cycle takes h_{j+1,j}
    from |A z|^2 - sum |h_kj|^2 and never forms the unused next basis vector.  Same solves, same
    iteration counts (+-1) as with the pass; every probe verified against its true residual."""
    for p in (p16, p128):
        n = p.A.shape[0]
        B = _rand((70, n), 52)
        out = {}
        try:
            for flag in (1, 0):
                p.eng.set_option("pyth_last", flag)
                X, iters, relres = p.mg.solve_batch(0, B, 1e-12)
                true_rel = np.linalg.norm(B.T - p.A @ X.T, axis=0) / np.linalg.norm(B.T, axis=0)
                assert true_rel.max() < 5e-12, (flag, true_rel.max())
                out[flag] = (X, np.asarray(iters))
        finally:
            p.eng.set_option("pyth_last", 1)
        assert np.abs(out[1][1] - out[0][1]).max() <= 1, (out[1][1], out[0][1])
        assert _relerr(out[1][0], out[0][0]) < 1e-9


def test_even_odd_reduced_outer_solve_equals_full_system_solve():
    """option eo_solve (default on, configurations whose stencil level is smoothed even-odd): the outer
    FGMRES works on the Schur complement of the even sites with half-length Krylov vectors (fgmres_eo).
    Against the full-system FGMRES (eo_solve = 0) on the same hierarchy: both reach a TRUE full-system
    residual below tol, the solutions agree to solver accuracy and with sparse LU, the iteration counts
    agree within +-1; a zero and a non-converging right-hand side behave the same; 70 columns = ragged batch."""
    from deflatedmlmc_schwinger_amd import hierarchy
    params = gateway.set_params('schwinger128')
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    mg = MG(A)
    mg.setup_solver_only(dict(hierarchy.TUNED_SOLVER_CFG_128))
    eng = mg.engine
    n = A.shape[0]
    B = _rand((70, n), 152)
    B[5] = 0.0
    B[9, n // 2:] = 0.0          # internal ordering differs, still a structured right-hand side
    out = {}
    try:
        for flag in (1, 0):
            eng.set_option("eo_solve", flag)
            X, iters, relres = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
            nrm = np.linalg.norm(B.T, axis=0)
            true_rel = np.linalg.norm(B.T - A @ X.T, axis=0) / np.where(nrm > 0, nrm, 1.0)
            assert true_rel.max() < 5e-12, (flag, true_rel.max())
            assert np.all(X[5] == 0) and iters[5] == 0
            out[flag] = (X, np.asarray(iters))
        # a budget of two iterations: reported, not raised, and not claimed converged
        eng.set_option("eo_solve", 1)
        X2, it2, rr2 = eng.solve(SOLVER_HID, 0, B[:3], 1e-12, 2)
        assert rr2.max() > 1e-12 and rr2.max() < 1.0
    finally:
        eng.set_option("eo_solve", 1)
    assert np.abs(out[1][1] - out[0][1]).max() <= 1, (out[1][1], out[0][1])
    assert _relerr(out[1][0], out[0][0]) < 1e-9
    lu = rp.LUSolver(A)
    ref = np.stack([lu(B[k]) for k in range(4)])
    assert _relerr(out[1][0][:4], ref) < 1e-8
    eng.close()


def test_block_level_solved_exactly_in_even_odd_reduced_form():
    """solver_cfg["direct_levels"] = [1]: the 4096-row block level carries the dense inverse of its even-odd
    Schur complement (even-odd operator 4, hierarchy.dense_schur_inverse_blocks) and is solved with it
    -- x_e = S^-1 (b_e - F b_o), x_o = G b_o - Hb x_e, four launches -- instead of smoothed; the levels
    below drop out of the cycle.  (i) operator 4 inverts operator 0 on the even rows (through the
    engine); (ii) solves reach the true residual and LU; (iii) not more iterations than with smoothing."""
    from deflatedmlmc_schwinger_amd import hierarchy
    params = gateway.set_params('schwinger128')
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    mg = MG(A)
    mg.setup_solver_only(dict(hierarchy.TUNED_SOLVER_CFG_128, direct_levels=[1]))
    eng = mg.engine
    Lc = 16
    site = np.arange(Lc * Lc)
    even = (((site % Lc) + (site // Lc)) & 1) == 0
    E = np.nonzero(np.repeat(even, 16))[0]
    X = np.zeros((3, Lc * Lc * 16), dtype=complex)
    X[:, E] = _rand((3, E.size), 77)
    Y = eng.apply_eo_operator(SOLVER_HID, 1, 0, X)
    Z = eng.apply_eo_operator(SOLVER_HID, 1, 4, Y)
    assert _relerr(Z[:, E], X[:, E]) < 1e-11
    n = A.shape[0]
    B = _rand((70, n), 153)
    out = {}
    try:
        for flag in (1, 0):
            eng.set_option("eo_direct", flag)
            Xs, iters, relres = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
            true_rel = np.linalg.norm(B.T - A @ Xs.T, axis=0) / np.linalg.norm(B.T, axis=0)
            assert true_rel.max() < 5e-12, (flag, true_rel.max())
            out[flag] = (Xs, np.asarray(iters))
    finally:
        eng.set_option("eo_direct", 1)
    print("iterations direct / smoothed:", out[1][1].max(), out[0][1].max())
    assert out[1][1].max() <= out[0][1].max()
    assert _relerr(out[1][0], out[0][0]) < 1e-9
    lu = rp.LUSolver(A)
    ref = np.stack([lu(B[k]) for k in range(4)])
    assert _relerr(out[1][0][:4], ref) < 1e-8
    eng.close()


def test_zero_rhs_and_single_rhs(p16):
    n = p16.A.shape[0]
    x, its, rr = p16.eng.solve(SOLVER_HID, 0, np.zeros(n, dtype=complex), 1e-12, 100)
    assert np.all(x == 0) and its == 0
    b = _rand(n, 60)
    p16.mg.level_nr = 0
    p16.mg.solve(p16.A, b, 1e-12)
    assert np.linalg.norm(b - p16.A @ p16.mg.x) / np.linalg.norm(b) < 5e-12
    assert p16.mg.num_iters >= 1


@pytest.mark.parametrize("prob,use_perm", [("p16", False), ("p128", True)])
def test_hutchinson_probes_match_lu_oracle(prob, use_perm, request):
    """per-probe e_k against the direct-LU oracle, 1e-10 relative (north star)."""
    p = request.getfixturevalue(prob)
    n = p.A.shape[0]
    np.random.seed(123456)
    probes = utils.draw_probes(12, n)
    ests, itf, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
    PT = p.levels[0].Pperm.transpose() if use_perm else None
    lu = p.lu_solver(0)
    for k in range(12):
        ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, p.Ux, PT)
        assert abs(ests[k] - ref) / abs(ref) < 1e-10, (k, ests[k], ref)
    assert itf.min() >= 1


def test_hutchinson_without_deflation(p16):
    n = p16.A.shape[0]
    np.random.seed(7)
    probes = utils.draw_probes(5, n)
    p16.eng.set_deflation(None)
    try:
        ests, _, _ = p16.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
    finally:
        p16.eng.set_deflation(np.asarray(p16.Ux))
    lu = p16.lu_solver(0)
    for k in range(5):
        ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, None, None)
        assert abs(ests[k] - ref) / abs(ref) < 1e-10


def test_mlmc_probes_match_lu_oracle(p128):
    """MLMC difference levels 0 (with level skipping) and 2 against exact level inverses."""
    p = p128
    cinv = np.asarray(p.mg.coarsest_inv)

    def solve_level(l, b):
        return p.lu_solver(l)(b)

    for level, mode, skip in ((0, MODE_MLMC_SKIP, True), (2, MODE_MLMC, False), (0, MODE_MLMC, False),
                              (1, MODE_MLMC, False)):
        n = p.levels[level].A.shape[0]
        np.random.seed(1000 + level)
        probes = utils.draw_probes(6, n)
        ests, itf, itc = p.eng.hutch_batch(mode, level, probes, 1e-12, 1000)
        scale = None
        for k in range(6):
            x0 = probes[k].astype(np.complex128)
            ref = rp.mlmc_probe(x0, level, p.levels, skip, solve_level, cinv, True)
            # differences of two O(100) numbers: tolerance relative to the minuend
            z = solve_level(level, p.levels[level].Bblock_perm @ (p.levels[level].Pperm.transpose() @ x0))
            scale = max(abs(np.vdot(x0, z)), abs(ref))
            assert abs(ests[k] - ref) / scale < 1e-10, (level, k, ests[k], ref)


def test_estimator_drop_in_hutchinson_16(p16, capsys):
    """the drop-in hutchinson() on 16^2 reproduces the sequential reference loop evaluated
    with exact (LU) solves on the same probe stream."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    lev0 = p16.levels[0]
    Ux, tr1, Vx, Sy = rp.deflation_hutchinson(p16.A, lev0.g3, None, 8, 1e-9, False)
    tp = dict(p16.tp)
    tp['nr_deflat_vctrs'] = 8
    tp['tol'] = 2.0e-2
    tp['batch'] = 64
    tp['deflation_eigenpairs'] = (Sy, Vx)     # same eigenpairs on both sides
    res = stoch_trace.hutchinson(p16.A, tp)
    capsys.readouterr()
    lu = p16.lu_solver(0)
    np.random.seed(123456)
    rough = [rp.hutch_probe(rp.rademacher(p16.A.shape[0]), lu, Ux, None) for _ in range(5)]
    rough_trace = np.sum(rough) / 5 + tr1
    tol = abs(tp['tol'] * rough_trace)
    ests = np.array([rp.hutch_probe(rp.rademacher(p16.A.shape[0]), lu, Ux, None)
                     for _ in range(res['nr_ests'] + 1)])
    idx, avg, dev = rp.stopping_rule(ests, tol)
    assert idx == res['nr_ests']
    assert abs((avg + tr1) - res['trace']) / abs(avg + tr1) < 1e-9
    assert abs(dev - res['std_dev']) / dev < 1e-8


def test_timers_and_launch_count(p16):
    p16.eng.set_profiling(True)
    p16.eng.timers_reset()
    B = _rand((2, p16.A.shape[0]), 70)
    p16.mg.solve_batch(0, B, 1e-10)
    t = p16.eng.timers()
    p16.eng.set_profiling(False)
    assert t["mvm"] > 0 and t["dots"] > 0 and t["axpy"] > 0
    assert p16.eng.launch_count() > 10


def test_bench_dirac_runs(p128):
    ms = p128.eng.bench_dirac(REF_HID, 0, 256, 5)
    assert 0 < ms < 50


EXACT_128 = -8.748242701374695 + 50.215154098005584j      # gateway.py:100-104


def _golden_rough_trace_128():
    """stoch_trace.py:288-301: mean of the first five deflated probes of seed 123456 + tr1, from
    the golden values the reference's own utils produced."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                    "golden.json")))
    first5 = np.array([complex(a, b) for a, b in g["hutch128_deflated_k8_seed123456"][:5]])
    return np.sum(first5) / 5 + complex(*g["defl128_tr1"])


def _check_level_rule(r, tol, rough, tol_fctr):
    """The stopping rule of stoch_trace.py:394-406 met ITS tolerance |tol * rough * tol_fctr| at
    the reported index and at no earlier index >= 5."""
    level_tol = abs(tol * rough * tol_fctr)
    assert abs(r['level_tol'] - level_tol) < 1e-6 * level_tol
    assert r['ests_dev'] / np.sqrt(r['nr_ests'] + 1) < level_tol
    idx, avg, dev = rp.stopping_rule(np.asarray(r['ests']), r['level_tol'])
    assert idx == r['nr_ests'] and len(r['ests']) == idx + 1
    assert dev == r['ests_dev']


def test_full_mlmc_flow_128_within_reported_error(capsys):
    """BASELINE config 3 / G202: the drop-in mlmc() on schwinger128 with the shipped preset
    (4 levels, level 1 skipped): every difference level stops exactly where the reference's rule
    stops for its tolerance share (sqrt(0.9), sqrt(0.1), stoch_trace.py:325-336,356-365), and the
    trace agrees with the reference's exact value within the estimator's own error."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    res = stoch_trace.mlmc(A, tp)
    capsys.readouterr()
    assert res['nr_levels'] == 4
    rough = _golden_rough_trace_128()
    assert abs(res['rough_trace'] - rough) < 1e-7 * abs(rough)
    _check_level_rule(res['results'][0], tp['tol'], rough, np.sqrt(0.9))
    _check_level_rule(res['results'][2], tp['tol'], rough, np.sqrt(0.1))
    err2 = 0.0
    for i in (0, 2):
        r = res['results'][i]
        assert r['nr_ests'] >= 5
        err2 += r['ests_dev'] ** 2 / (r['nr_ests'] + 1)
    assert res['results'][1]['nr_ests'] == 0            # skipped level
    assert res['results'][3]['nr_ests'] == 1            # direct coarsest term
    err = np.sqrt(err2)
    # the rule bounds the combined error by |tol * rough| (0.9 + 0.1 of its square)
    assert err < abs(tp['tol'] * rough)
    assert abs(res['trace'] - EXACT_128) < 4.0 * err + 1e-9, (res['trace'], err)
    assert res['total_complexity'] > 0


def test_full_mlmc_flow_128_without_level_skipping(capsys):
    """The non-skip 4-level branch (mlmc_levels_to_skip = []; tolerance fractions 0.45 / 0.45 /
    0.1, stoch_trace.py:325-336): difference levels 0, 1 and 2 plus the direct coarsest term."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    params['mlmc_levels_to_skip'] = []
    params['trace_tol'] = 3.0e-2
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    res = stoch_trace.mlmc(A, tp)
    capsys.readouterr()
    rough = _golden_rough_trace_128()
    err2 = 0.0
    for i, fctr in ((0, np.sqrt(0.45)), (1, np.sqrt(0.45)), (2, np.sqrt(0.1))):
        r = res['results'][i]
        _check_level_rule(r, tp['tol'], rough, fctr)
        assert r['function_iters'] > r['nr_ests']
        err2 += r['ests_dev'] ** 2 / (r['nr_ests'] + 1)
    assert res['results'][3]['nr_ests'] == 1
    err = np.sqrt(err2)
    assert err < abs(tp['tol'] * rough)
    assert abs(res['trace'] - EXACT_128) < 4.0 * err + 1e-9, (res['trace'], err)


def test_mlmc_inexact_01_deflation_16(capsys):
    """defl_type = "inexact_01" (utils.py:177-182): V need not be eigenvectors, the low-rank part
    tr1 = tr(V^H D V) is computed with the difference operator itself (engine solves), and
    tr(D (I - V V^H)) + tr1 = tr(D) for ANY orthonormal V -- checked against exact LU here."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    from scipy.sparse.linalg import LinearOperator
    p = Problem('schwinger16', 8)
    tp = dict(p.tp)
    tp['defl_type'] = 'inexact_01'
    tp['defl_eigvs_tol_MLMC'] = 1.0e-2          # deliberately rough vectors
    tp['diff_lev_op_tol'] = 1.0e-3
    cinv = np.asarray(p.mg.coarsest_inv)
    mg = p.mg
    mg.skip_level = False
    for lvl in (0, 1):
        mg.level_for_diff_op = lvl
        n = p.levels[lvl].A.shape[0]
        lop = LinearOperator((n, n), dtype=np.complex128,
                             matvec=lambda v: mg.diff_op_Q(np.array(v, dtype=np.complex128)))
        Vx, Ux, tr1 = utils.deflation_pre_computations(p.A, 4, tp['defl_eigvs_tol_MLMC'], "mlmc",
                                                       mg.timer, tp, mg, lop, level_nr=lvl)
        capsys.readouterr()
        # oracle: tr(V^H (A_f^-1 - P A_c^-1 R) V) with exact solves
        lev = p.levels[lvl]
        lc = lvl + 1
        ref = 0.0
        for i in range(4):
            v = Vx[:, i]
            t1 = p.lu_solver(lvl)(v)
            vc = lev.R @ v
            t2 = cinv @ vc if lc == len(p.levels) - 1 else p.lu_solver(lc)(vc)
            ref += np.vdot(v, t1 - lev.P @ np.asarray(t2).reshape(-1))
        assert abs(tr1 - ref) < 1e-9 * max(1.0, abs(ref)), (lvl, tr1, ref)
        # probes with the projection registered by deflation_pre_computations
        np.random.seed(900 + lvl)
        probes = utils.draw_probes(4, n)
        ests, _, _ = p.eng.hutch_batch(MODE_MLMC, lvl, probes, 1e-12, 1000)
        for k in range(4):
            refk = rp.mlmc_probe(probes[k].astype(np.complex128), lvl, p.levels, False,
                                 lambda l, b: p.lu_solver(l)(b), cinv, False, Vx=Vx)
            assert abs(ests[k] - refk) < 1e-9 * max(1.0, abs(refk))
        p.eng.set_level_deflation(lvl, None)
    # the whole flow: unbiased with inexact vectors
    params = gateway.set_params('schwinger16')
    params.update({'function_tol': 1e-12, 'nr_deflat_vctrs': 8, 'mlmc_deflat_vctrs': [4, 4],
                   'defl_type': 'inexact_01', 'defl_eigvs_tol_MLMC': 1.0e-2,
                   'trace_tol': 2.0e-2, 'mlmc_levels_to_skip': []})
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tpf = utils.trace_params_from_params(params, "mlmc")
    tpf['batch'] = 64
    res = stoch_trace.mlmc(A, tpf)
    capsys.readouterr()
    exact = 265.8581064657958
    err2 = sum(res['results'][i]['ests_dev'] ** 2 / (res['results'][i]['nr_ests'] + 1) for i in (0, 1))
    assert abs(res['trace'] - exact) < 4.0 * np.sqrt(err2) + 1e-6 * exact, (res['trace'], np.sqrt(err2))


def test_full_deflated_hutchinson_flow_128_within_reported_error(capsys):
    """G102: deflated Hutchinson, looser tolerance so the run stays short."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    params['trace_tol'] = 5.0e-2
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "hutchinson")
    res = stoch_trace.hutchinson(A, tp)
    capsys.readouterr()
    n = res['nr_ests'] + 1
    err = res['std_dev'] / np.sqrt(n)
    assert abs(res['trace'] - EXACT_128) < 4.0 * err
    assert res['function_iters'] >= n


@pytest.mark.parametrize("L", [8, 32, 64])
def test_synthetic_lattices_stencil_and_solve(L):
    """random U(1) gauge fields at other lattice sizes (towards BASELINE config 5): stencil vs
    CSR, batched solve vs LU, plain Hutchinson probes vs LU -- without any multigrid levels
    (single-level hierarchy, unpreconditioned GMRES) and, for L = 64, with the solver hierarchy."""
    from deflatedmlmc_schwinger_amd.engine import Engine
    from deflatedmlmc_schwinger_amd import hierarchy
    mass = 0.05
    A = matrix.synthetic_matrix(L, mass, sigma=0.3, seed=100 + L)
    n = A.shape[0]
    lat = hierarchy.detect_lattice(A)
    assert lat is not None and lat[0] == L
    eng = Engine(0)
    eng.hier_begin(0, 1)
    eng.set_lattice(0, L, lat[1], lat[2], lat[3])
    eng.hier_end(0)
    eng.set_solver(32, 0)
    X = _rand((5, n), L)
    assert _relerr(eng.apply_dirac(0, 0, X), (A @ X.T).T) < 1e-13
    lu = rp.LUSolver(A)
    B = _rand((3, n), L + 1)
    Xs, its, rr = eng.solve(0, 0, B, 1e-11, 2000)
    assert rr.max() < 1e-11
    ref = np.stack([lu(b) for b in B])
    assert _relerr(Xs, ref) < 1e-8
    np.random.seed(L)
    probes = utils.draw_probes(4, n)
    ests, _, _ = eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 4000)
    for k in range(4):
        refe = rp.hutch_probe(probes[k].astype(np.complex128), lu, None, None)
        assert abs(ests[k] - refe) / abs(refe) < 1e-9
    eng.close()
    if L == 64:
        mg = MG(A)
        tp = {'use_permuted': False, 'test_vectors_type': 'EVs', 'latt_dims': [L, L],
              'x_displacement': 0, 'solver_cfg': dict(hierarchy.DEFAULT_SOLVER_CFG)}
        mg.setup(dof=[2, 8, 8], aggrs=[16, 4], max_levels=3, dim=2, acc_eigvs='high',
                 sys_type='schwinger', params=tp)
        Xs, its, rr = mg.solve_batch(0, B, 1e-12)
        assert rr.max() < 1e-12 and its.max() < 60
        assert _relerr(Xs, ref) < 1e-9


def test_mlmc_level_deflation_projection_matches_oracle(p16):
    """probe body with MLMC-level deflation vectors registered (utils.py:260-266)."""
    p = p16
    cinv = np.asarray(p.mg.coarsest_inv)
    rng = np.random.default_rng(11)
    for level in (0, 1):
        n = p.levels[level].A.shape[0]
        V, _ = np.linalg.qr(rng.standard_normal((n, 5)) + 1j * rng.standard_normal((n, 5)))
        p.eng.set_level_deflation(level, V)
        try:
            np.random.seed(300 + level)
            probes = utils.draw_probes(4, n)
            ests, _, _ = p.eng.hutch_batch(MODE_MLMC, level, probes, 1e-12, 1000)
        finally:
            p.eng.set_level_deflation(level, None)
        for k in range(4):
            ref = rp.mlmc_probe(probes[k].astype(np.complex128), level, p.levels, False,
                                lambda l, b: p.lu_solver(l)(b), cinv, False, Vx=V)
            assert abs(ests[k] - ref) < 1e-9 * max(1.0, abs(ref))


def test_full_deflated_mlmc_flow_16(capsys):
    """G201-like: MLMC with deflation of the difference operators (mlmc_deflat_vctrs > 0) on
    16^2; eigenvectors computed tightly so the estimator is unbiased to the test's resolution."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    params = gateway.set_params('schwinger16')
    params['function_tol'] = 1e-12
    params['nr_deflat_vctrs'] = 8
    params['mlmc_deflat_vctrs'] = [8, 8]
    params['defl_eigvs_tol_MLMC'] = 1.0e-8
    params['diff_lev_op_tol'] = 1.0e-11
    params['trace_tol'] = 2.0e-2
    params['accuracy_mg_eigvs'] = 'high'
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    tp['batch'] = 64
    res = stoch_trace.mlmc(A, tp)
    capsys.readouterr()
    exact = 265.8581064657958
    err2 = sum(res['results'][i]['ests_dev'] ** 2 / (res['results'][i]['nr_ests'] + 1) for i in (0,))
    assert abs(res['trace'] - exact) < 4.0 * np.sqrt(err2) + 1e-6 * exact, (res['trace'], np.sqrt(err2))


def test_synthetic_three_level_mg_probes_match_lu():
    """BASELINE config 5 in miniature: synthetic random-gauge lattice, 3-level solver hierarchy
    with a K-cycle, plain Hutchinson probes as one batch, against sparse LU.  (128^2 keeps the
    host-side ARPACK/SuperLU setup short; the same test passes at 256^2 in ~3.5 min.)"""
    from deflatedmlmc_schwinger_amd import hierarchy
    L, mass = 128, -0.02
    A = matrix.synthetic_matrix(L, mass, sigma=0.35, seed=2024)
    n = A.shape[0]
    mg = MG(A)
    tp = {'use_permuted': False, 'test_vectors_type': 'EVs', 'latt_dims': [L, L],
          'x_displacement': 0,
          'solver_cfg': dict(hierarchy.DEFAULT_SOLVER_CFG, coarsening=[(4, 8), (4, 8)],
                             cycle=[(0, 7, 2), (0, 7, 0)])}
    # reference-style hierarchy with 2 coarsenings (strip aggregates) + the solver hierarchy
    mg.setup(dof=[2, 8, 8], aggrs=[16, 4], max_levels=3, dim=2, acc_eigvs='high',
             sys_type='schwinger', params=tp)
    assert mg.solver_info["levels"] == [n, n // 2, n // 32]
    lu = rp.LUSolver(A)
    np.random.seed(2024)
    probes = utils.draw_probes(6, n)
    mg.engine.set_deflation(None)
    ests, itf, _ = mg.engine.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
    assert itf.max() < 80
    for k in range(6):
        ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, None, None)
        assert abs(ests[k] - ref) / abs(ref) < 1e-10, (k, ests[k], ref)


def test_nonconvergence_is_reported_not_raised(p128):
    """the reference discards fgmres' exitCode (multigrid.py:362): running out of iterations is not
    an error here either, it shows in relres / iters."""
    B = _rand((2, p128.A.shape[0]), 80)
    X, iters, relres = p128.eng.solve(SOLVER_HID, 0, B, 1e-12, 3)
    assert relres.min() > 1e-12 and np.all(np.isfinite(X))
    assert iters.max() <= 3
    true_rel = np.linalg.norm(B.T - p128.A @ X.T, axis=0) / np.linalg.norm(B.T, axis=0)
    assert true_rel.max() < 1.0          # three iterations still reduce the residual


def test_deflation_rank_64_on_16(p16):
    """the 16^2 preset deflates 64 vectors (gateway.py:81); the engine projects in chunks of 32."""
    n = p16.A.shape[0]
    lev0 = p16.levels[0]
    Ux, tr1, _, _ = rp.deflation_hutchinson(p16.A, lev0.g3, None, 64, 1e-9, False)
    p16.eng.set_deflation(np.asarray(Ux))
    try:
        np.random.seed(64)
        probes = utils.draw_probes(5, n)
        ests, _, _ = p16.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
    finally:
        p16.eng.set_deflation(np.asarray(p16.Ux))
    lu = p16.lu_solver(0)
    for k in range(5):
        ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, Ux, None)
        assert abs(ests[k] - ref) < 1e-10 * max(abs(ref), 1.0)


def test_argument_errors_surface_as_messages(p16):
    from deflatedmlmc_schwinger_amd.engine import EngineError
    n = p16.A.shape[0]
    with pytest.raises(EngineError, match="vector length"):
        p16.eng.apply_dirac(REF_HID, 0, np.zeros(n + 1, dtype=complex))
    with pytest.raises(EngineError, match="level"):
        p16.eng._chk(p16.eng._lib.sw_apply_dirac(p16.eng._h, 0, 9, 1, None, None), "sw_apply_dirac")
    with pytest.raises(EngineError, match="bad arguments"):
        p16.eng._chk(p16.eng._lib.sw_solve(p16.eng._h, 0, 0, 0, None, None, 1e-12, 10, None, None),
                     "sw_solve")
    with pytest.raises(EngineError, match="restart"):
        p16.eng.set_solver(1000, 0)
    with pytest.raises(EngineError, match="no probes uploaded|Hutchinson mode runs at level 0"):
        p16.eng.hutch_run(MODE_HUTCHINSON, 1, 1e-12, 10)


def test_adaptive_gpu_setup_gives_a_working_hierarchy(p128):
    """solver hierarchy from engine-side inverse iteration (no ARPACK / SuperLU):
    same solve quality as the eigenvector-based one, per-probe parity unchanged."""
    from deflatedmlmc_schwinger_amd import hierarchy
    mg = p128.mg
    base_iters = None
    B = _rand((4, p128.A.shape[0]), 90)
    X0, it0, _ = mg.solve_batch(0, B, 1e-12)
    cfg = dict(hierarchy.DEFAULT_SOLVER_CFG, setup="adaptive", setup_sweeps=3, setup_tol=1e-1,
               setup_maxiter=100)
    try:
        mg.upload_solver_hierarchy(cfg)
        assert mg.solver_info["setup_log"] is not None
        X1, it1, rr = mg.solve_batch(0, B, 1e-12)
        assert rr.max() < 1e-12
        assert it1.max() <= it0.max() + 6
        assert _relerr(X1, X0) < 1e-8
        np.random.seed(123456)
        probes = utils.draw_probes(4, p128.A.shape[0])
        ests, _, _ = mg.engine.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
        lu = p128.lu_solver(0)
        for k in range(4):
            ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, p128.Ux,
                                 p128.levels[0].Pperm.transpose())
            assert abs(ests[k] - ref) / abs(ref) < 1e-10
    finally:
        mg.upload_solver_hierarchy(None)


def test_config4_probe_stream_statistics_and_sharding_independence(p128):
    """BASELINE config 4 on one GPU: the first 4096 probes of the MT19937(123456) stream.
    (i) the deflated-Hutchinson mean agrees with the exact trace within the estimator's own
    error; (ii) a probe's estimate does not depend on the batch it was solved in (what makes
    contiguous sharding over ranks legitimate) -- same values to 1e-10 for 256- and 64-probe
    batches taken from the same stream positions."""
    from deflatedmlmc_schwinger_amd.engine import ProbeStream
    n = p128.A.shape[0]
    ps = ProbeStream(123456)
    ests = []
    first = None
    for b in range(16):
        probes = ps.rademacher(256, n)
        if b == 3:
            first = probes
        e, _, _ = p128.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
        ests.append(e)
    ests = np.concatenate(ests)
    mean = ests.mean() + p128.tr1
    err = ests.std() / np.sqrt(ests.size)
    assert abs(mean - EXACT_128) < 4.0 * err, (mean, err)
    # rank r of 4 would take probes [r*64, (r+1)*64) of that round
    e_round = ests[3 * 256:4 * 256]
    for r in (0, 3):
        e_shard, _, _ = p128.eng.hutch_batch(MODE_HUTCHINSON, 0, first[r * 64:(r + 1) * 64], 1e-12, 1000)
        ref = e_round[r * 64:(r + 1) * 64]
        assert np.max(np.abs(e_shard - ref) / np.abs(ref)) < 1e-10


def test_z4_probes_build_only_option(p16):
    """BASELINE config 1 asks for Z4 probes; the reference only has Z2 (utils.py:213-216), so Z4
    is a flagged build-only option: entries {1,i,-1,-i}, e = x^H A^-1 x against the LU oracle,
    32 probes on 16^2, and the mean against the exact trace within the estimator's error."""
    n = p16.A.shape[0]
    np.random.seed(2468)
    codes = utils.draw_probes(32, n, "z4")
    assert set(np.unique(codes).tolist()) == {-2, -1, 1, 2}
    X = utils.probes_as_complex(codes)
    assert np.allclose(np.abs(X), 1.0)
    p16.eng.set_deflation(None)
    try:
        ests, _, _ = p16.eng.hutch_batch(MODE_HUTCHINSON, 0, codes, 1e-12, 1000)
    finally:
        p16.eng.set_deflation(np.asarray(p16.Ux))
    lu = p16.lu_solver(0)
    for k in range(32):
        ref = np.vdot(X[k], lu(X[k]))
        assert abs(ests[k] - ref) / abs(ref) < 1e-10
    exact = 265.8581064657958
    assert abs(ests.mean() - exact) < 4.0 * ests.std() / np.sqrt(32)


def test_stochastic_coarsest_level_build_only_option(p16, p128, capsys):
    """the coarsest-level term estimated stochastically (the reference raises
    "Stochastic coarsest-level computation is disabled", stoch_trace.py:436-437; here it is a flagged
    build-only option).  Per probe e = x^H A_c^-1 (Bblock_perm Pperm^T) x against NumPy, the mean
    against the direct value np.trace(Pperm^H A_c^-1 Bblock_perm) of stoch_trace.py:431-435."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    for p in (p16, p128):
        last = len(p.levels) - 1
        cinv = np.asarray(p.mg.coarsest_inv)
        n = cinv.shape[0]
        np.random.seed(77)
        probes = utils.draw_probes(40, n)
        ests, _, _ = p.eng.hutch_batch(MODE_LEVEL, last, probes, 1e-12, 1000)
        lev = p.levels[last]
        for k in range(40):
            x = probes[k].astype(np.complex128)
            rhs = x if isinstance(lev.Pperm, int) else lev.Bblock_perm @ (lev.Pperm.transpose() @ x)
            ref = np.vdot(x, cinv @ rhs)
            assert abs(ests[k] - ref) < 1e-11 * max(1.0, abs(ref))
    # whole flow on 16^2: default raises like the reference, the flag runs the estimator
    params = gateway.set_params('schwinger16')
    params.update({'function_tol': 1e-12, 'nr_deflat_vctrs': 8, 'mlmc_deflat_vctrs': [0, 0],
                   'trace_tol': 2.0e-2, 'coarsest_level_directly': False})
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    tp['batch'] = 64
    with pytest.raises(Exception, match="Stochastic coarsest-level computation is disabled"):
        stoch_trace.mlmc(A, tp)
    capsys.readouterr()
    tp['stochastic_coarsest'] = True
    res = stoch_trace.mlmc(A, tp)
    capsys.readouterr()
    exact = 265.8581064657958
    last = res['nr_levels'] - 1
    assert res['results'][last]['nr_ests'] >= 5
    err2 = sum(res['results'][i]['ests_dev'] ** 2 / (res['results'][i]['nr_ests'] + 1)
               for i in (0, last))
    assert abs(res['trace'] - exact) < 4.0 * np.sqrt(err2) + 1e-6 * exact, (res['trace'], np.sqrt(err2))


def test_reference_faithful_cycle_reports_reference_iteration_counts():
    """with ref_smoother="gmres30x2" level-0 solves are preconditioned by
    the REFERENCE hierarchy and MG.one_mg_step's own cycle (multigrid.py:369-447) with the smoother the
    reference calls, lgmres(maxiter=2) = GMRES(30) followed by an LGMRES cycle of 30 Arnoldi steps
    augmented with the first cycle's correction (engine option lgmres_aug), so `function_iters` is the
    reference's count.  Checked against the oracle's restatement (SciPy's own lgmres as the smoother +
    flexible GMRES): same outer iteration count (+-1; measured: identical on all five probes), same
    solution."""
    for name, nprobe in (('schwinger16', 4), ('schwinger128', 1)):
        params = gateway.set_params(name)
        params['function_tol'] = 1e-12
        params['use_solver_hierarchy'] = False
        params['ref_smoother'] = 'gmres30x2'
        params['solver_restart'] = 32
        A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
        tp = utils.trace_params_from_params(params, "hutchinson")
        mg = MG(A)
        mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
                 acc_eigvs=tp['accuracy_mg_eigvs'], sys_type='schwinger', params=tp)
        omg = rp.OracleMG(A)
        omg.setup(tp['dof'], tp['aggrs'], tp['max_nr_levels'], tp['accuracy_mg_eigvs'], tp,
                  testvectors=mg.testvectors)
        np.random.seed(123456)
        B = utils.probes_as_complex(utils.draw_probes(nprobe, A.shape[0]))
        X, its, relres = mg.solve_batch(0, B, 1e-12)
        X, its = np.atleast_2d(X), np.atleast_1d(its)
        for k in range(nprobe):
            omg.level_nr = 0
            omg.solve(A, B[k], 1e-12)
            print("reference-faithful cycle, %s probe %d: engine %d outer iterations, oracle (SciPy lgmres "
                  "smoother) %d" % (name, k, int(its[k]), int(omg.num_iters)))
            assert abs(int(its[k]) - int(omg.num_iters)) <= 1, (name, k, its[k], omg.num_iters)
            assert _relerr(X[k], omg.x) < 1e-9
        if name == 'schwinger128':
            assert 10 <= int(its[0]) <= 16          # 13 outer iterations per plain probe
        mg.engine.close()


def test_device_side_setup_builds_an_equivalent_hierarchy(p128):
    """SURVEY 8f-2 on the device: test vectors, per-aggregate QR, P / R and the Galerkin products
    all built by engine kernels (solver_cfg["setup"] = "device").  Checked: P^H P = I and
    A_c = R A P through the C ABI, per-probe parity against LU at 1e-10, and an outer iteration
    count no worse than the ARPACK-based hierarchy's by more than a few."""
    p = p128
    n = p.A.shape[0]
    np.random.seed(4321)
    probes = utils.draw_probes(16, n)
    ests_ref, its_ref, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
    from deflatedmlmc_schwinger_amd import hierarchy
    cfg = dict(hierarchy.DEFAULT_SOLVER_CFG, setup="device")
    try:
        p.mg.upload_solver_hierarchy(cfg)
        info = p.mg.solver_info
        assert info["levels"] == [32768, 16384, 4096]
        eng = p.eng
        for lvl in (0, 1):
            nc = info["levels"][lvl + 1]
            Xc = _rand((3, nc), 50 + lvl)
            PX = eng.prolong(SOLVER_HID, lvl, Xc)
            assert _relerr(eng.restrict(SOLVER_HID, lvl, PX), Xc) < 1e-12          # R P = I
            APX = eng.apply_dirac(SOLVER_HID, lvl, PX)
            assert _relerr(eng.apply_dirac(SOLVER_HID, lvl + 1, Xc),
                           eng.restrict(SOLVER_HID, lvl, APX)) < 1e-12              # A_c = R A P
        ests, its, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
        lu = p.lu_solver(0)
        PT = p.levels[0].Pperm.transpose()
        for k in range(16):
            ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, p.Ux, PT)
            assert abs(ests[k] - ref) / abs(ref) < 1e-10
        assert int(its.max()) <= int(its_ref.max()) + 4, (its.max(), its_ref.max())
        for entry in info["setup_log"]:
            if entry.get("pass") == 2:
                assert max(entry["gmres_iterations"]) < 64            # converged, not capped
    finally:
        p.mg.upload_solver_hierarchy(None)


def test_c_abi_collective_single_rank(p16):
    """sw_comm_* / sw_allreduce_stats (RCCL behind the C ABI): on the one GPU of the
    test box a one-rank communicator; the reduction must return the statistics unchanged and the
    derived mean / population deviation must be those of stoch_trace.py:143-145."""
    from deflatedmlmc_schwinger_amd import dist
    from deflatedmlmc_schwinger_amd.engine import Engine
    uid = Engine.comm_unique_id()
    assert len(uid) == 128
    p16.eng.comm_init(1, 0, uid)
    e = _rand(37, 9)
    stats = dist.local_stats(e)
    out = p16.eng.allreduce_stats(stats)
    assert np.array_equal(out, stats)
    mean, std = dist.mean_and_population_std(out)
    assert abs(mean - e.mean()) < 1e-13 and abs(std - np.sqrt(np.mean(np.abs(e - e.mean()) ** 2))) < 1e-12
    p16.eng.comm_destroy()
    with pytest.raises(Exception, match="no communicator"):
        p16.eng.allreduce_stats(stats)


def test_mixed_convergence_after_a_stale_sync_hint(p128):
    """ADVICE r1 (lazy sync): after a batch that left a large iteration hint, a batch mixing a
    right-hand side that converges at once (b = A P y: the coarse correction is exact for it), a
    zero right-hand side and ordinary ones.  Converged probes are frozen (no division by a
    round-off h_{j+1,j}); every solution must meet the TRUE residual bound and equal LU."""
    p = p128
    n = p.A.shape[0]
    hard = _rand((64, n), 123)
    _, its_hard, _ = p.mg.solve_batch(0, hard, 1e-12)           # leaves sync_hint ~ 14
    assert int(np.max(its_hard)) >= 8
    P0, P1 = p.mg.solver_hier["P"][0], p.mg.solver_hier["P"][1]
    rng = np.random.default_rng(77)
    y = rng.standard_normal(P1.shape[1]) + 1j * rng.standard_normal(P1.shape[1])
    B = np.zeros((5, n), dtype=np.complex128)
    # x = P0 P1 y lies in the range of both prolongators: every coarse correction of the cycle is
    # exact for b = A x (dense inverse at the bottom), so these two converge at once
    B[0] = p.A @ (P0 @ (P1 @ y))
    B[1] = _rand((n,), 5)
    B[3] = 1e-9 * _rand((n,), 6)                                # tiny but non-zero
    B[4] = p.A @ (P0 @ (P1 @ y[::-1].copy()))
    X, its, relres = p.mg.solve_batch(0, B, 1e-12)              # B[2] = 0
    assert its[2] == 0 and np.all(X[2] == 0)
    assert its[0] <= 3 and its[4] <= 3 and its[1] >= its[0] + 4
    lu = p.lu_solver(0)
    for k in (0, 1, 3, 4):
        r = B[k] - p.A @ X[k]
        assert np.linalg.norm(r) < 2e-12 * np.linalg.norm(B[k])
        assert _relerr(X[k], lu(B[k])) < 1e-9
        assert relres[k] < 1e-12


def test_redefining_hierarchy_zero_resets_probe_state():
    """sw_hier_begin(hid 0) drops everything sized by the previous definition
    (deflation vectors, permutations, probe slots and workspace), so a handle can move from a
    larger lattice to a smaller one and back."""
    from deflatedmlmc_schwinger_amd import hierarchy
    from deflatedmlmc_schwinger_amd.engine import Engine
    eng = Engine(0)
    for L in (32, 8, 16):
        A = matrix.synthetic_matrix(L, 0.2, 0.3, 5)
        lat = hierarchy.detect_lattice(A)
        eng.hier_begin(REF_HID, 1)
        eng.set_lattice(REF_HID, lat[0], lat[1], lat[2], lat[3])
        eng.hier_end(REF_HID)
        eng.set_solver(16, REF_HID)
        n = A.shape[0]
        U, _ = np.linalg.qr(_rand((n, 3), L))
        eng.set_deflation(U)
        np.random.seed(L)
        probes = utils.draw_probes(5, n)
        ests, _, _ = eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
        lu = rp.LUSolver(A)
        for k in range(5):
            ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, U, None)
            assert abs(ests[k] - ref) < 1e-10 * max(1.0, abs(ref))
    eng.close()


def test_even_odd_schur_smoother_matches_model_and_solves(p16, p128):
    """sw_set_eo_smoother (k_eo_hop / k_schur_step): the level-0 cycle with the even-odd
    post-smoother against the NumPy model built from the explicit Schur complement, and converged
    solves through it against LU (per-probe parity 1e-10 on 128^2)."""
    from deflatedmlmc_schwinger_amd import hierarchy
    for p, L in ((p16, 16), (p128, 128)):
        for nu in (1, 4):
            cfg = dict(hierarchy.DEFAULT_SOLVER_CFG, cycle=[(0, nu, 0), (0, 7, 0)], eo_smoother=True)
            try:
                p.mg.upload_solver_hierarchy(cfg, testvectors=p.mg.solver_testvectors)
                sh = p.mg.solver_hier
                S, E, O, D = hierarchy.schur_complement(sh["A"][0], L)
                B = _rand((sh["A"][0].shape[0], 3), 91)
                ref = em.cycle_eo(sh["A"], sh["P"], sh["coarsest_inv"], [tuple(c) for c in cfg["cycle"]],
                                  B, p.mg.solver_weights, p.mg.solver_weights_eo, E, O, D)
                X = p.eng.vcycle(SOLVER_HID, 0, B.T.copy())
                assert _relerr(X.T, ref) < 1e-10, (L, nu)
                if nu == 4:
                    n = p.A.shape[0]
                    np.random.seed(2718)
                    probes = utils.draw_probes(6, n)
                    ests, its, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
                    lu = p.lu_solver(0)
                    PT = None if isinstance(p.levels[0].Pperm, int) else p.levels[0].Pperm.transpose()
                    for k in range(6):
                        ref_e = rp.hutch_probe(probes[k].astype(np.complex128), lu, p.Ux, PT)
                        assert abs(ests[k] - ref_e) / abs(ref_e) < 1e-10
                    assert int(its.max()) < 40
            finally:
                p.mg.upload_solver_hierarchy(None, testvectors=p.mg.solver_testvectors)


def test_even_odd_smoother_on_block_levels_matches_model(p128):
    """sw_set_eo_operator: the even-odd Schur smoother on the coarse (block) levels -- S, F = A_eo
    D_oo^-1, G = D_oo^-1, Hb = D_oo^-1 A_oe as subset operators on the MFMA block-row kernel --
    against the NumPy model, on a 4-level hierarchy with all three fine levels smoothed even-odd;
    converged solves through it against LU."""
    from deflatedmlmc_schwinger_amd import hierarchy
    p = p128
    cfg = dict(hierarchy.DEFAULT_SOLVER_CFG, coarsening=[(4, 8), (2, 8), (2, 8)],
               cycle=[(0, 4, 0), (0, 3, 0), (0, 5, 0)], eo_levels=[0, 1, 2])
    try:
        p.mg.upload_solver_hierarchy(cfg)
        sh = p.mg.solver_hier
        assert sorted(p.mg.solver_eo) == [0, 1, 2]
        cyc = [tuple(c) for c in cfg["cycle"]]
        for level0 in (2, 1, 0):
            B = _rand((sh["A"][level0].shape[0], 3), 17 + level0)
            ref = em.cycle(sh["A"], sh["P"], sh["coarsest_inv"], cyc, level0, B,
                           weights=p.mg.solver_weights, eo=p.mg.solver_eo)
            X = p.eng.vcycle(SOLVER_HID, level0, B.T.copy())
            assert _relerr(X.T, ref) < 1e-9, level0
        n = p.A.shape[0]
        np.random.seed(99)
        probes = utils.draw_probes(6, n)
        ests, its, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
        lu = p.lu_solver(0)
        PT = p.levels[0].Pperm.transpose()
        for k in range(6):
            ref_e = rp.hutch_probe(probes[k].astype(np.complex128), lu, p.Ux, PT)
            assert abs(ests[k] - ref_e) / abs(ref_e) < 1e-10
        assert int(its.max()) < 40
    finally:
        p.mg.upload_solver_hierarchy(None)


@pytest.mark.parametrize("kcycle", [0, 2])
def test_single_precision_preconditioner_keeps_fp64_results(p128, kcycle):
    """Option precond_f32 (cfg key precond_precision = "f32"): the multigrid cycle inside the fp64
    flexible GMRES runs in complex64 (k_bsr_mfma_f32, k_schur_step<float2>, k_ell<.., float2>).
    The cycle itself equals the fp64 cycle to single-precision round-off; the converged solves are fp64:
    per-probe estimates against the sparse-LU oracle at the north-star tolerance 1e-10, with the
    iteration count of the fp64 preconditioner (+-2).  With a K-cycle the small inner FGMRES stays fp64
    and only its preconditioner is complex64."""
    from deflatedmlmc_schwinger_amd import hierarchy
    p = p128
    base = dict(hierarchy.DEFAULT_SOLVER_CFG, coarsening=[(4, 8), (2, 8), (2, 8)],
                cycle=[(0, 4, 0), (0, 3, kcycle), (0, 5, 0)], eo_levels=[0, 1, 2])
    assert hierarchy.f32_capable(base)
    n = p.A.shape[0]
    np.random.seed(4242)
    probes = utils.draw_probes(70, n)
    lu = p.lu_solver(0)
    PT = p.levels[0].Pperm.transpose()
    try:
        p.mg.upload_solver_hierarchy(base)
        tvs = p.mg.solver_testvectors
        X64 = {}
        for level0 in (2, 1, 0):
            B = _rand((p.mg.solver_hier["A"][level0].shape[0], 70), 23 + level0)
            X64[level0] = (B, p.eng.vcycle(SOLVER_HID, level0, B.T.copy()))
        _, its64, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
        p.mg.upload_solver_hierarchy(dict(base, precond_precision="f32"), testvectors=tvs)
        for level0 in (2, 1, 0):
            B, ref = X64[level0]
            X = p.eng.vcycle(SOLVER_HID, level0, B.T.copy())
            err = _relerr(X, ref)
            assert 1e-9 < err < 2e-5, (level0, err)      # single precision, and really single precision
        # f32_krylov = 1 (default): the restart cycles keep their Krylov basis in complex64 as well
        # (iterative refinement: residual and solution update in fp64 once per restart); 0: fp64 basis
        for fk in (1, 0):
            p.eng.set_option("f32_krylov", fk)
            ests, its, _ = p.eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
            for k in range(0, 70, 7):
                ref_e = rp.hutch_probe(probes[k].astype(np.complex128), lu, p.Ux, PT)
                assert abs(ests[k] - ref_e) / abs(ref_e) < 1e-10, (fk, k)
            assert abs(int(its.max()) - int(its64.max())) <= 2, (fk, its.max(), its64.max())
    finally:
        p.eng.set_option("f32_krylov", 1)
        p.eng.set_option("precond_f32", 0)
        p.mg.upload_solver_hierarchy(None)


def test_even_odd_operators_built_on_the_device_match_the_host_construction():
    """sw_setup_eo_operators (k_block_inverse / k_block_products: G = D_oo^-1, F = A_eo G, Hb = G A_oe,
    S = D_ee - F A_oe by batched 16 x 16 algebra on the device) against hierarchy.coarse_schur_blocks on
    the operator read back from the same engine level, through sw_apply_eo_operator on random vectors;
    then the device-built hierarchy with every level smoothed even-odd solves to LU parity."""
    from deflatedmlmc_schwinger_amd import hierarchy
    L = 128
    A = matrix.synthetic_matrix(L, -0.02, sigma=0.3, seed=77)
    cfg = dict(hierarchy.TUNED_SOLVER_CFG_128, coarsening=[(4, 8), (2, 8), (2, 8)],
               cycle=[(0, 4, 0), (0, 3, 0), (0, 5, 0)], eo_levels=[0, 1, 2])
    mg = MG(A)
    mg.setup_solver_only(cfg)
    eng = mg.engine
    assert mg.solver_info["levels"] == [2 * L * L, L * L, L * L // 4, L * L // 16]
    for level, Lc in ((1, 32), (2, 16)):
        nbr, blk = hierarchy.site_blocks_from_block_rows(*eng.level_bsr(SOLVER_HID, level))
        ops = hierarchy.coarse_schur_blocks(nbr, blk, Lc)
        n = Lc * Lc * 16
        X = _rand((3, n), 400 + level)
        for which, (tmap, kcol, vals) in enumerate(ops["packed"]):
            RT, KS = kcol.shape
            r = np.repeat(tmap.astype(np.int64) * 16, KS * 64).reshape(RT, KS, 4, 16) + np.arange(16)
            c = kcol.astype(np.int64)[:, :, None, None] + np.arange(4)[None, None, :, None] + \
                np.zeros((1, 1, 1, 16), int)
            M = sp.csr_matrix((vals.reshape(RT, KS, 4, 16).ravel(), (r.ravel(), c.ravel())), shape=(n, n))
            Y = eng.apply_eo_operator(SOLVER_HID, level, which, X)
            assert _relerr(Y, (M @ X.T).T) < 1e-12, (level, which)
    lu = rp.LUSolver(A)
    np.random.seed(5)
    probes = utils.draw_probes(5, A.shape[0])
    ests, its, _ = eng.hutch_batch(MODE_HUTCHINSON, 0, probes, 1e-12, 1000)
    for k in range(5):
        ref = rp.hutch_probe(probes[k].astype(np.complex128), lu, None, None)
        assert abs(ests[k] - ref) / abs(ref) < 1e-10
    assert int(its.max()) < 40
