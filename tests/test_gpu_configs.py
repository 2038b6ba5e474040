"""GPU tests of the BASELINE configurations at their written sizes and of the round-3 engine paths:

* config 5 at full size: the synthetic 1024^2 lattice with bench.py's own five-level device-built
  hierarchy (size-independent properties: host-recomputed true residuals through the CSR operator,
  estimates recomputed from the returned solutions, bit-exact probe codes);
* config 3 as written: 2-level MLMC difference A_0^-1 - P A_c^-1 R on 32768 -> 8192 with the dense
  8192^2 coarse solve, per probe against the oracle's restatement of utils.py:252-361 with LU solves;
* hierarchies built from singular-vector test vectors (multigrid.py:159-188) on the GPU;
* the strict per-probe parity mode (engine option stop_factor), the in-launch reductions
  (fused_reduce) and the device-side setup pieces (Arnoldi for the smoother polynomials, the dense
  Schur inverse of a directly solved block level).

Tolerances: floating point complex128; per-probe estimates 1e-10 relative (north star), differences of
two O(100) numbers relative to the minuend, operator applications 1e-12."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from deflatedmlmc_schwinger_amd import gateway, hierarchy, matrix, utils  # noqa: E402
from deflatedmlmc_schwinger_amd.engine import (MODE_HUTCHINSON, MODE_MLMC, MODE_MLMC_SKIP,  # noqa: E402
                                               ProbeStream)
from deflatedmlmc_schwinger_amd.multigrid import MG, REF_HID, SOLVER_HID  # noqa: E402
from oracle import ref_path as rp  # noqa: E402


def _rand(shape, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def _relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


@pytest.mark.parametrize("nlevels,nb", [(5, 128), (3, 128)])
def test_config5_full_size_synthetic_1024_lattice(nlevels, nb):
    """BASELINE config 5's lattice at its written size: synthetic 1024^2 random U(1) gauge
    field (seed 2024, sigma 0.204: mean plaquette ~0.92; m = -0.05), 2 097 152 unknowns, at the batch width
    of bench.py's records (128 probes), through (5) the five-level hierarchy bench.py --workload synthetic
    uses and (3) config 5 AS WRITTEN, a three-level hierarchy 2 097 152 -> 262 144 -> 4 096
    (hierarchy.synthetic_solver_cfg(L, levels=3)), both built on the device.  No LU at this size; instead
    (i) the probe codes generated on the device are np.random's legacy stream bit for bit,
    (ii) the TRUE residual of every returned solution, recomputed on the host with the CSR operator
        assembled from the links, is below 5e-12 (tol 1e-12 on the engine's own residual),
    (iii) the estimates of the probe driver equal x^H z recomputed on the host from the solutions."""
    L, mass = 1024, -0.05
    U1, U2 = matrix.synthetic_links(L, 0.204, 2024)
    cfg = hierarchy.synthetic_solver_cfg(L, levels=3 if nlevels == 3 else None)
    mg = MG((L, mass, U1, U2))
    mg.setup_solver_only(cfg, device=0, engines=1)
    assert mg.solver_info["levels"] == ([2097152, 262144, 4096] if nlevels == 3 else
                                        [2097152, 262144, 65536, 16384, 4096])
    eng = mg.engine
    n = 2 * L * L
    eng.stream_set(ProbeStream(123456).window())
    eng.probes_generate(0, 0, nb, 0)
    eng.probes_select(0)
    eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000)
    ests, itf, _ = eng.hutch_fetch()
    codes = eng.probes_fetch(0)
    np.random.seed(123456)
    ref_codes = (2 * np.random.randint(2, size=(nb, n)) - 1).astype(np.int8)
    assert np.array_equal(codes, ref_codes)                                  # (i)
    del ref_codes
    assert 1 <= itf.min() and itf.max() <= (20 if nlevels == 5 else 40), (itf.min(), itf.max())
    B = codes.astype(np.complex128)
    X, its, relres = eng.solve(SOLVER_HID, 0, B, 1e-12, 1000)
    assert relres.max() < 1e-12
    A = (hierarchy.wilson_from_links(U1, U2, L)
         + mass * sp.identity(n, dtype=np.complex128, format="csr")).tocsr()
    worst = 0.0
    for k0 in range(0, nb, 8):                                               # (ii), 8 columns at a time
        R = B[k0:k0 + 8].T - A @ X[k0:k0 + 8].T
        worst = max(worst, float(np.max(np.linalg.norm(R, axis=0) / np.linalg.norm(B[k0:k0 + 8].T, axis=0))))
    print("1024^2, %d levels, %d probes: worst true relative residual %.2e, iterations %d..%d"
          % (nlevels, nb, worst, its.min(), its.max()))
    assert worst < 5e-12
    e_host = np.einsum("kn,kn->k", B, X)                                     # (iii): probes are real
    assert np.max(np.abs(e_host - ests) / np.abs(e_host)) < 1e-10
    # linearity of the stencil at this size (a size-independent property of the operator kernel)
    Z = _rand((2, n), 5)
    lhs = eng.apply_dirac(SOLVER_HID, 0, (0.3 - 0.2j) * Z[0] + Z[1])
    assert _relerr(lhs, A @ ((0.3 - 0.2j) * Z[0] + Z[1])) < 1e-13
    eng.close()


def test_wide_batch_on_the_512_lattice_walks_the_strips_chunk_by_chunk():
    """A 128-probe batch on the synthetic 512^2 lattice: strips of the full batch width that fit the cache
    would be 32 lattice rows, fewer than the 10 smoother launches reach (36), so the time-skewed schedule is
    walked once per 64-probe chunk with strips of 64 rows (skew_height).  The solutions must be bit-identical
    to the plain order's, and their TRUE residuals, recomputed on the host with the CSR operator, below 5e-12."""
    L, mass, nb = 512, -0.05, 128
    U1, U2 = matrix.synthetic_links(L, 0.204, 2024)
    mg = MG((L, mass, U1, U2))
    mg.setup_solver_only(hierarchy.synthetic_solver_cfg(L), device=0, engines=1)
    eng = mg.engine
    n = 2 * L * L
    B = _rand((nb, n), 512)
    out = {}
    try:
        for H in (-1, 0):
            eng.set_option("eo_skew", H)
            eng.timers_reset()
            X, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 1000)
            out[H] = (X, np.asarray(its), eng.launch_count())
    finally:
        eng.set_option("eo_skew", -1)
    assert out[-1][2] > out[0][2]                                   # the strips really ran
    assert np.array_equal(out[-1][0], out[0][0]) and np.array_equal(out[-1][1], out[0][1])
    A = (hierarchy.wilson_from_links(U1, U2, L) + mass * sp.identity(n, dtype=np.complex128, format="csr")).tocsr()
    worst = 0.0
    for k0 in range(0, nb, 16):
        R = B[k0:k0 + 16].T - A @ out[-1][0][k0:k0 + 16].T
        worst = max(worst, float(np.max(np.linalg.norm(R, axis=0) / np.linalg.norm(B[k0:k0 + 16].T, axis=0))))
    assert worst < 5e-12, worst
    eng.close()


@pytest.mark.parametrize("coarsest", ["dense", "eo"])
def test_config3_as_written_two_level_mlmc_difference(coarsest):
    """BASELINE config 3 literally: schwinger128, a 2-level hierarchy 32768 -> 8192 (the
    reference's aggregation), the MLMC difference probe e = x^H A_0^-1 C x - x^H P A_c^-1 R C x with the
    coarse solve done by the dense 8192^2 inverse on the matrix cores -- or (build key ref_coarsest = "eo")
    exactly in even-odd form with the dense 4096^2 inverse of the coarse level's Schur complement --, against
    the oracle's restatement of utils.py:252-361 with sparse-LU solves -- and the direct coarse term
    tr(Pperm^H A_c^-1 Bblock_perm) (stoch_trace.py:428-435) so that difference + coarse term reproduce the
    exact trace."""
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    params['ref_coarsest'] = coarsest
    params.update({'max_nr_levels': 2, 'nr_deflat_vctrs': 0})
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    tp['mlmc_deflat_vctrs'] = [0]
    mg = MG(A)
    mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=2, dim=2, acc_eigvs=tp['accuracy_mg_eigvs'],
             sys_type='schwinger', params=tp)
    levels = mg.ml.levels
    assert [lev.A.shape[0] for lev in levels] == [32768, 8192]
    assert (mg.coarse_eo is not None) == (coarsest == "eo")
    cinv = np.asarray(mg.coarsest_inv)
    lu = rp.LUSolver(A)
    n = A.shape[0]
    np.random.seed(2468)
    probes = utils.draw_probes(8, n)
    ests, itf, itc = mg.engine.hutch_batch(MODE_MLMC, 0, probes, 1e-12, 1000)
    assert np.all(itc == 1)                          # the coarse "solve" is the dense inverse
    for k in range(8):
        x0 = probes[k].astype(np.complex128)
        ref = rp.mlmc_probe(x0, 0, levels, False, lambda l, b: lu(b), cinv, True)
        z = lu(levels[0].Bblock_perm @ (levels[0].Pperm.transpose() @ x0))
        scale = max(abs(np.vdot(x0, z)), abs(ref))
        assert abs(ests[k] - ref) / scale < 1e-10, (k, ests[k], ref)
    # The coarse term (stoch_trace.py:428-435) and the telescoping identity of SURVEY 3.1,
    #   tr[(A_0^-1 - P A_c^-1 R) M_0] + tr[A_c^-1 R M_0 P] = tr[A_0^-1 M_0]  (= gateway.py:104),
    # with B_1 M_1 = R M_0 P: the mean of 512 difference probes from the GPU plus the direct coarse term
    # reproduces the reference's "exact trace" comment within the estimator's own error.
    coarse = np.trace(levels[1].Pperm.transpose().conjugate() @ (cinv @ levels[1].Bblock_perm))
    G = (levels[0].R @ (levels[0].Pperm.transpose() @ levels[0].P)).toarray()
    assert abs(coarse - np.sum(cinv * G.T)) < 1e-8 * abs(coarse)
    np.random.seed(123456)
    big = utils.draw_probes(512, n)
    e1, _, _ = mg.engine.hutch_batch(MODE_MLMC, 0, big[:256], 1e-12, 1000)
    e2, _, _ = mg.engine.hutch_batch(MODE_MLMC, 0, big[256:], 1e-12, 1000)
    e = np.concatenate([e1, e2])
    err = np.std(e) / np.sqrt(e.size)
    total = np.mean(e) + coarse
    print("config 3: difference %.4f%+.4fj +- %.3f, coarse %.4f%+.4fj, sum %.4f%+.4fj"
          % (np.mean(e).real, np.mean(e).imag, err, coarse.real, coarse.imag, total.real, total.imag))
    assert abs(total - gateway.EXACT_TRACE_SCHWINGER128) < 4.0 * err
    mg.engine.close()


@pytest.mark.parametrize("name,tv_type", [("schwinger16", "LSVs"), ("schwinger16", "RSVs"),
                                          ("schwinger128", "LSVs"), ("schwinger128", "RSVs")])
def test_singular_vector_hierarchies_run_on_the_gpu(name, tv_type):
    """hierarchies built from singular-vector test vectors (test_vectors_type 'LSVs' /
    'RSVs', multigrid.py:159-188) uploaded to the engine: eight MLMC level-0 difference probes each
    against the oracle's probe body with LU solves on the same hierarchy, 1e-10."""
    params = gateway.set_params(name)
    params['function_tol'] = 1e-12
    params['test_vectors_type'] = tv_type
    params['accuracy_mg_eigvs'] = 'high'
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    tp['mlmc_deflat_vctrs'] = [0] * 3
    mg = MG(A)
    mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
             acc_eigvs='high', sys_type='schwinger', params=tp)
    levels = mg.ml.levels
    nlev = len(levels)
    assert nlev >= 3
    cinv = np.asarray(mg.coarsest_inv)
    lus = {}

    def solve_level(l, b):
        if l not in lus:
            lus[l] = rp.LUSolver(levels[l].A)
        return lus[l](b)

    n = A.shape[0]
    use_perm = bool(tp['use_permuted'])
    skip = nlev >= 4
    np.random.seed(97)
    probes = utils.draw_probes(8, n)
    mode = MODE_MLMC_SKIP if skip else MODE_MLMC
    ests, itf, itc = mg.engine.hutch_batch(mode, 0, probes, 1e-12, 1000)
    for k in range(8):
        x0 = probes[k].astype(np.complex128)
        ref = rp.mlmc_probe(x0, 0, levels, skip, solve_level, cinv, use_perm)
        xd = levels[0].Bblock_perm @ (levels[0].Pperm.transpose() @ x0) if use_perm else x0
        scale = max(abs(np.vdot(x0, solve_level(0, xd))), abs(ref))
        assert abs(ests[k] - ref) / scale < 1e-10, (name, tv_type, k, ests[k], ref)
    mg.engine.close()


def _tuned128(engines=1, extra=None):
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    params['solver_cfg'] = dict(hierarchy.TUNED_SOLVER_CFG_128, **(extra or {}))
    params['engines'] = engines
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "hutchinson")
    mg = MG(A)
    mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
             acc_eigvs=tp['accuracy_mg_eigvs'], sys_type='schwinger', params=tp)
    return A, tp, mg


def test_in_launch_reductions_match_the_two_stage_ones():
    """engine option fused_reduce: the inner products of the Krylov solver completed inside the launch
    that forms their partial sums (last block done, two levels), the FGMRES scalar updates riding along.
    Same solves as with the separate reduction kernels -- same iteration counts, solutions equal to
    round-off (the order of the partial sums differs between the two) -- and bit-identical results when
    repeated (the order of every sum is fixed, whichever workgroup does the adding)."""
    A, tp, mg = _tuned128()
    eng = mg.engine
    n = A.shape[0]
    B = _rand((100, n), 31)                      # two probe chunks, one of them ragged
    out = {}
    try:
        for flag in (1, 0, 1):
            eng.set_option("fused_reduce", flag)
            for hid in (SOLVER_HID, REF_HID):    # even-odd reduced FGMRES and the full-system one
                X, its, rr = eng.solve(hid, 0, B[:100 if hid == SOLVER_HID else 3], 1e-12, 400)
                assert rr.max() < 1e-12
                out.setdefault((flag, hid), []).append((X, np.asarray(its)))
    finally:
        eng.set_option("fused_reduce", 1)
    for hid in (SOLVER_HID, REF_HID):
        (Xa, ia), (Xb, ib) = out[(1, hid)]
        assert np.array_equal(Xa, Xb) and np.array_equal(ia, ib)            # deterministic
        Xc, ic = out[(0, hid)][0]
        assert np.max(np.abs(ia - ic)) <= 1
        assert _relerr(Xa, Xc) < 1e-10
    # launch count: a batch must need fewer launches with the reductions folded in
    eng.timers_reset()
    eng.solve(SOLVER_HID, 0, B[:64], 1e-12, 400)
    fused = eng.launch_count()
    eng.set_option("fused_reduce", 0)
    eng.timers_reset()
    eng.solve(SOLVER_HID, 0, B[:64], 1e-12, 400)
    plain = eng.launch_count()
    eng.set_option("fused_reduce", 1)
    print("launches per solve: fused %d, two-stage %d" % (fused, plain))
    assert fused < plain
    eng.close()


def test_strict_parity_mode_iterates_below_the_reference_tolerance():
    """engine option stop_factor = 0.1: the batch is iterated until every TRUE residual is below
    0.1 * tol while the reported iteration counts stay those at tol (the reference's count,
    multigrid.py:347-366).  Cost: at most one more outer iteration for the batch."""
    A, tp, mg = _tuned128()
    eng = mg.engine
    n = A.shape[0]
    B = _rand((64, n), 77)
    X1, it1, rr1 = eng.solve(SOLVER_HID, 0, B, 1e-12, 400)
    eng.timers_reset()
    try:
        eng.set_option("stop_factor", 0.1)
        X2, it2, rr2 = eng.solve(SOLVER_HID, 0, B, 1e-12, 400)
    finally:
        eng.set_option("stop_factor", 1.0)
    t1 = np.linalg.norm(B.T - A @ X1.T, axis=0) / np.linalg.norm(B.T, axis=0)
    t2 = np.linalg.norm(B.T - A @ X2.T, axis=0) / np.linalg.norm(B.T, axis=0)
    print("true residuals: default %.2e, strict %.2e; reported iterations %d / %d"
          % (t1.max(), t2.max(), it1.max(), it2.max()))
    assert t1.max() < 5e-12 and t2.max() < 1.5e-13
    assert rr2.max() < 1e-13
    assert np.all(np.abs(np.asarray(it2) - np.asarray(it1)) <= 1)          # counts are taken at tol
    eng.close()


def test_device_arnoldi_gives_a_smoother_polynomial_as_good_as_the_hosts():
    """sw_setup_arnoldi (f2): the Arnoldi relation behind the smoother polynomial computed on the device.
    The random start vectors differ from the host's, so the weights are not compared entry by entry;
    checked instead: (i) the Ritz values lie in the operator's numerical range (16^2: dense Hermitian parts),
    (ii) a polynomial of the device's weights damps a random residual as well as the host-fitted one
    (within a factor 2), for the level operator and for the even-odd Schur complement."""
    params = gateway.set_params('schwinger16')
    params['function_tol'] = 1e-12
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "hutchinson")
    tp['solver_cfg'] = dict(hierarchy.DEFAULT_SOLVER_CFG)
    mg = MG(A)
    mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
             acc_eigvs='high', sys_type='schwinger', params=tp)
    eng = mg.engine
    L = 16
    S, E, O, D = hierarchy.schur_complement(A, L)
    # the stencil level must carry an even-odd smoother for which = 1 to be defined there
    for which, M in ((0, sp.csr_matrix(A)), (1, S)):
        deg = 8
        H = eng.setup_arnoldi(SOLVER_HID, 0, which, deg)
        assert H.shape == (deg + 1, deg)
        assert np.allclose(np.tril(H[:deg, :], -2), 0.0)                      # Hessenberg
        ritz = np.linalg.eigvals(H[:deg, :])
        Md = M.toarray()
        hr = np.linalg.eigvalsh(0.5 * (Md + Md.conj().T))                    # numerical range: real extent
        hi = np.linalg.eigvalsh(-0.5j * (Md - Md.conj().T))                   # ... imaginary extent
        assert hr[0] - 1e-9 < ritz.real.min() and ritz.real.max() < hr[-1] + 1e-9
        assert hi[0] - 1e-9 < ritz.imag.min() and ritz.imag.max() < hi[-1] + 1e-9
        w_dev = hierarchy.weights_from_hessenberg(H)
        w_host = hierarchy.smoother_weights(M, deg)
        r0 = _rand(M.shape[0], 12)

        def damp(w):
            x = np.zeros_like(r0)
            for wk in w:
                x = x + wk * (r0 - M @ x)
            return np.linalg.norm(r0 - M @ x) / np.linalg.norm(r0)
        dd, dh = damp(w_dev), damp(w_host)
        print("which %d: residual after %d steps -- device weights %.3e, host weights %.3e" % (which, deg, dd, dh))
        assert dd < 2.0 * dh and dd < 1.0
    eng.close()


def test_dense_schur_inverse_built_on_the_device_matches_the_host_route():
    """sw_setup_direct_level (f2): the dense inverse of the 4096-row level's even-odd Schur complement
    formed on the device (S -> dense -> Gauss-Jordan -> block rows) against the host LAPACK route
    (hierarchy.dense_schur_inverse_blocks), applied through the C ABI."""
    params = gateway.set_params('schwinger128')
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    Lc = 16
    site = np.arange(Lc * Lc)
    even = (((site % Lc) + (site // Lc)) & 1) == 0
    E = np.nonzero(np.repeat(even, 16))[0]
    X = np.zeros((5, Lc * Lc * 16), dtype=complex)
    X[:, E] = _rand((5, E.size), 8)
    res = {}
    for route in ("device", "host"):
        mg = MG(A)
        mg.setup_solver_only(dict(hierarchy.TUNED_SOLVER_CFG_128, direct_levels=[1], setup_direct=route))
        res[route] = mg.engine.apply_eo_operator(SOLVER_HID, 1, 4, X)
        if route == "device":
            Y = mg.engine.apply_eo_operator(SOLVER_HID, 1, 0, res[route])
            assert _relerr(Y[:, E], X[:, E]) < 1e-11                        # S S^-1 = 1 on the even rows
        mg.engine.close()
    assert _relerr(res["device"][:, E], res["host"][:, E]) < 1e-10
    assert np.abs(np.delete(res["device"], E, axis=1)).max() == 0.0          # odd rows untouched


def test_blocked_gauss_jordan_inverse_matches_the_unblocked_one():
    """engine option gj_block: the device-side dense inverse with the pivot steps of a 32-column panel
    confined to the panel and the rest of the matrix updated once per panel on the matrix cores, against the
    unblocked algorithm (a rank-1 update of the whole matrix per pivot step): the 1024-row coarsest operator
    of the tuned hierarchy and the 2048-row Schur complement of its 4096-row level, panels of 32 and 64
    columns -- both inverses applied through the C ABI, compared with each other and checked as inverses
    (S S^-1 = 1; A_c A_c^-1 = 1 through the level's block-row operator)."""
    A, tp, mg = _tuned128(extra={"direct_levels": [1]})
    eng = mg.engine
    nc = mg.solver_info["levels"][-1]
    last = len(mg.solver_info["levels"]) - 1
    Lc = 16
    site = np.arange(Lc * Lc)
    even = (((site % Lc) + (site // Lc)) & 1) == 0
    E = np.nonzero(np.repeat(even, 16))[0]
    Xs = np.zeros((5, Lc * Lc * 16), dtype=complex)
    Xs[:, E] = _rand((5, E.size), 81)
    Xc = _rand((5, nc), 82)
    out = {}
    try:
        for blk in (0, 32, 64):
            eng.set_option("gj_block", blk)
            eng.setup_invert_coarsest(SOLVER_HID)
            eng.setup_direct_level(SOLVER_HID, 1)
            Yc = eng.coarsest(SOLVER_HID, Xc)
            Ys = eng.apply_eo_operator(SOLVER_HID, 1, 4, Xs)
            assert _relerr(eng.apply_dirac(SOLVER_HID, last, Yc), Xc) < 1e-11, blk
            assert _relerr(eng.apply_eo_operator(SOLVER_HID, 1, 0, Ys)[:, E], Xs[:, E]) < 1e-11, blk
            out[blk] = (Yc, Ys)
    finally:
        eng.set_option("gj_block", 32)
    for blk in (32, 64):
        assert _relerr(out[blk][0], out[0][0]) < 1e-11
        assert _relerr(out[blk][1], out[0][1]) < 1e-11
    eng.close()


def test_three_product_block_row_kernel_matches_the_four_product_one():
    """engine option mfma_3m: the block-row operator with three real matrix products per complex one
    (k_bsr_mfma3: T1 = Ar Xr, T2 = Ai Xi, T3 = (Ar + Ai)(Xr + Xi)) against the four-product kernel and
    against NumPy -- level operator (modes Y = A X and the smoother step inside a cycle), the dense
    coarsest inverse, the even-odd subset operators incl. the dense Schur inverse, for one, two and four
    tiles of 16 probes per wave and a ragged batch."""
    A, tp, mg = _tuned128(extra={"direct_levels": [1]})
    eng = mg.engine
    n1 = mg.solver_info["levels"][1]
    nc = mg.solver_info["levels"][-1]
    Lc = 16
    site = np.arange(Lc * Lc)
    even = (((site % Lc) + (site // Lc)) & 1) == 0
    E = np.nonzero(np.repeat(even, 16))[0]
    for nb in (3, 70):
        X1 = _rand((nb, n1), 3 + nb)
        Xc = _rand((nb, nc), 4 + nb)
        Xe = np.zeros((nb, n1), dtype=complex)
        Xe[:, E] = _rand((nb, E.size), 5 + nb)
        B0 = _rand((nb, A.shape[0]), 6 + nb)
        ref = None
        try:
            for m3, tiles in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 4)):
                eng.set_option("mfma_3m", m3)
                eng.set_option("mfma3_tiles", tiles)
                got = (eng.apply_dirac(SOLVER_HID, 1, X1), eng.coarsest(SOLVER_HID, Xc),
                       eng.apply_eo_operator(SOLVER_HID, 1, 0, Xe), eng.apply_eo_operator(SOLVER_HID, 1, 4, Xe),
                       eng.apply_eo_operator(SOLVER_HID, 1, 3, Xe), eng.vcycle(SOLVER_HID, 0, B0))
                if ref is None:
                    ref = got
                    continue
                for a, b, tol in zip(got, ref, (1e-13, 1e-12, 1e-13, 1e-11, 1e-13, 1e-11)):
                    assert _relerr(a, b) < tol, (nb, m3, tiles, _relerr(a, b))
        finally:
            eng.set_option("mfma_3m", 1)
            eng.set_option("mfma3_tiles", 0)
    # and against the operator itself, assembled on the host from the engine's block rows
    kcol, vals = eng.level_bsr(SOLVER_HID, 1)
    A1 = hierarchy.matrix_from_block_rows(kcol, vals, n1)
    X1 = _rand((5, n1), 99)
    assert _relerr(eng.apply_dirac(SOLVER_HID, 1, X1), (A1 @ X1.T).T) < 1e-13
    eng.close()


def test_time_skewed_smoother_order_is_bit_identical():
    """engine option eo_skew: the Schur steps of the lattice level's even-odd smoother applied strip by
    strip in a time-skewed order (shrinking trapezoid, parallelograms, closing wedge on the periodic
    lattice; what keeps the smoother's working set inside the Infinity Cache on lattices beyond it).
    Same arithmetic per site in another order: the cycle and the solves must be BIT-identical to the
    plain order -- strips of 32 and of 64 lattice rows on the 128^2 lattice (four and two strips), an odd
    and an even number of steps (the ping-pong parity differs), full-system cycle and reduced-system solve."""
    for nu in (8, 7):
        A, tp, mg = _tuned128(extra={"cycle": [(0, nu, 0), (0, 10, 0)]})
        eng = mg.engine
        n = A.shape[0]
        B = _rand((70, n), 19 + nu)
        ref = None
        try:
            for H, chunked in ((0, 0), (32, 0), (64, 0), (32, 1)):
                eng.set_option("eo_skew", H)
                eng.set_option("eo_skew_chunk", chunked)      # strips walked 64-probe chunk by chunk
                eng.timers_reset()
                Xc = eng.vcycle(SOLVER_HID, 0, B)                           # full-system cycle (eo_smooth)
                launches = eng.launch_count()
                Xs, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)       # reduced system (vcycle_even)
                if ref is None:
                    ref = (Xc, Xs, np.asarray(its), launches)
                    continue
                assert launches > ref[3]                                    # the strips really ran
                assert np.array_equal(Xc, ref[0]), (nu, H)
                assert np.array_equal(Xs, ref[1]) and np.array_equal(np.asarray(its), ref[2]), (nu, H)
        finally:
            eng.set_option("eo_skew", -1)
            eng.set_option("eo_skew_chunk", 0)
        eng.close()


def test_lds_tiled_schur_kernel_is_bit_identical():
    """engine option eo_tile: the fp64 even-odd launches of the lattice level that carry no b' operand (the
    reduced system's operator S x and the product-form factors v - u S v) from LDS-staged halo tiles in rotated
    coordinates, double-buffered by LDS-DMA with the link values gathered into LDS alongside (k_schur_tile, 4 and
    8 waves per persistent workgroup), against the one-wave-per-item k_schur_step: the same arithmetic per
    site in the same order, so cycles and solves must be BIT-identical -- ragged batch (70 probes: two chunks),
    256 probes (four chunks: eight jobs per CU), the 128^2 and a 16^2 lattice (tiles wrap around the torus)."""
    A, tp, mg = _tuned128()
    eng = mg.engine
    n = A.shape[0]
    try:
        for nb, seed in ((70, 77), (256, 78)):
            B = _rand((nb, n), seed)
            ref = None
            for tile in (0, 4, 8):
                eng.set_option("eo_tile", tile)
                eng.timers_reset()
                Xc = eng.vcycle(SOLVER_HID, 0, B)
                Xs, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
                Sx = eng.apply_dirac(SOLVER_HID, 0, B[:3])
                if tile == 0:
                    ref = (Xc, Xs, np.asarray(its), Sx)
                    continue
                assert np.array_equal(Xc, ref[0]), (nb, tile)
                assert np.array_equal(Xs, ref[1]) and np.array_equal(np.asarray(its), ref[2]), (nb, tile)
                assert np.array_equal(Sx, ref[3])
    finally:
        eng.set_option("eo_tile", 0)
    eng.close()


def test_lds_staged_dense_kernel_is_bit_identical():
    """engine option dense_lds (default on): dense operators -- the dense Schur inverse of the directly solved
    4096-row level (2048^2, through tmap / kcol: the even sites' tiles), the 1024^2 coarsest inverse -- applied by
    k_dense_mfma3_lds (operands staged through LDS by a ring of LDS-DMA fills, 64-row x 32-probe blocks per
    workgroup) against k_bsr_mfma3 (fragments straight from L2): the same three-product arithmetic over the same
    k-steps in the same order, so cycles, coarsest applications and solves must be BIT-identical; 70 probes (two
    64-probe chunks, i.e. four probe pairs) and 256."""
    A, tp, mg = _tuned128()
    eng = mg.engine
    n = A.shape[0]
    nc = mg.solver_info["levels"][-1]
    try:
        for nb, seed in ((70, 91), (256, 92)):
            B = _rand((nb, n), seed)
            C = _rand((nb, nc), seed + 10)
            ref = None
            for on in (0, 1):
                eng.set_option("dense_lds", on)
                Xc = eng.vcycle(SOLVER_HID, 0, B)
                Yc = eng.coarsest(SOLVER_HID, C)
                Xs, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
                if ref is None:
                    ref = (Xc, Yc, Xs, np.asarray(its))
                    continue
                assert np.array_equal(Yc, ref[1]), nb
                assert np.array_equal(Xc, ref[0]), nb
                assert np.array_equal(Xs, ref[2]) and np.array_equal(np.asarray(its), ref[3]), nb
    finally:
        eng.set_option("dense_lds", 1)
    eng.close()


def test_product_form_smoother_matches_the_step_form():
    """engine option eo_product: the even-odd smoother of the reduced-system cycle as
    x + beta prod_j (1 - u_j S) (b' - S x) -- the factors read one half vector and write one, 2 nu + 2 passes
    instead of the 3 nu of the steps x <- x + w_k (b' - S x) -- against the step form: the same polynomial, so
    the same solves to round-off (identical iteration counts, solutions equal to 1e-11, true residuals below
    tol); plain order and time-skewed strips bit-identical to each other; 70 probes (two chunks), an even and
    an odd number of steps; and on a 48^2 lattice (extent not a power of two) built on the device."""
    from deflatedmlmc_schwinger_amd import hierarchy as swhier, matrix as swmatrix
    from deflatedmlmc_schwinger_amd.multigrid import MG
    for nu in (8, 7):
        A, tp, mg = _tuned128(extra={"cycle": [(0, nu, 0), (0, 10, 0)]})
        eng = mg.engine
        n = A.shape[0]
        B = _rand((70, n), 23 + nu)
        out = {}
        try:
            for prod, H in ((0, 0), (1, 0), (1, 32), (1, 64)):
                eng.set_option("eo_product", prod)
                eng.set_option("eo_skew", H)
                eng.timers_reset()
                Xs, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
                out[(prod, H)] = (Xs, np.asarray(its), eng.launch_count())
                true = np.linalg.norm(B.T - A @ Xs.T, axis=0) / np.linalg.norm(B.T, axis=0)
                assert true.max() < 5e-12, (nu, prod, H, true.max())
        finally:
            eng.set_option("eo_skew", -1)
            eng.set_option("eo_product", 1)
        assert np.array_equal(out[(1, 0)][1], out[(0, 0)][1])                  # same iteration counts
        assert out[(1, 0)][2] == out[(0, 0)][2]                                # and launches
        assert _relerr(out[(1, 0)][0], out[(0, 0)][0]) < 1e-11
        for H in (32, 64):
            assert out[(1, H)][2] > out[(1, 0)][2]                             # the strips really ran
            assert np.array_equal(out[(1, H)][0], out[(1, 0)][0]), (nu, H)
            assert np.array_equal(out[(1, H)][1], out[(1, 0)][1]), (nu, H)
        eng.close()
    L = 48
    U1, U2 = swmatrix.synthetic_links(L, 0.204, 7)
    mg = MG((L, -0.05, U1, U2))
    cfg = swhier.synthetic_solver_cfg(L, 4, "device")      # 4 steps: strips of 16 rows are admissible
    mg.setup_solver_only(cfg, device=0, engines=1)
    eng = mg.engine
    n = 2 * L * L
    B = _rand((64, n), 77)
    out = {}
    try:
        for prod, H in ((0, 0), (1, 0), (1, 16)):
            eng.set_option("eo_product", prod)
            eng.set_option("eo_skew", H)
            Xs, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
            assert np.max(rr) < 1e-12
            out[(prod, H)] = (Xs, np.asarray(its))
    finally:
        eng.set_option("eo_skew", -1)
        eng.set_option("eo_product", 1)
    assert np.array_equal(out[(1, 0)][1], out[(0, 0)][1]) and _relerr(out[(1, 0)][0], out[(0, 0)][0]) < 1e-11
    assert np.array_equal(out[(1, 16)][0], out[(1, 0)][0])
    eng.close()


def test_gram_matrix_restart_cycles_equal_the_arnoldi_ones():
    """engine option gram_cycle: the restart cycles of the even-odd reduced outer solve in Gram-matrix form
    (directions built without orthogonalisation, ONE pass for all inner products, per-probe Cholesky solve of
    the normal equations) against the Arnoldi / Givens form: same Krylov space, so the same solutions to
    solver accuracy, true residuals below tol, per-probe iteration counts within one; mixed batches (zero
    right-hand sides, an early-converging one, a ragged width), the strict stopping mode, and a batch that
    must not converge within its iteration budget."""
    A, tp, mg = _tuned128()
    eng = mg.engine
    n = A.shape[0]
    B = _rand((70, n), 4711)
    B[3] = 0.0                                    # zero right-hand side
    B[5] = A @ _rand(n, 1) * 1e-3                 # ordinary, small norm
    out = {}
    try:
        for flag in (1, 0):
            eng.set_option("gram_cycle", flag)
            for rep in range(2):                  # second solve: with the sync hint of the first
                X, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
            nrm = np.linalg.norm(B.T, axis=0)
            true = np.linalg.norm(B.T - A @ X.T, axis=0) / np.where(nrm > 0, nrm, 1.0)
            assert true.max() < 5e-12, (flag, true.max())
            assert np.all(X[3] == 0.0) and its[3] == 0
            out[flag] = (X, np.asarray(its))
        assert _relerr(out[1][0], out[0][0]) < 1e-10
        assert np.max(np.abs(out[1][1] - out[0][1])) <= 1, (out[1][1], out[0][1])
        eng.set_option("gram_cycle", 1)
        eng.set_option("stop_factor", 0.1)
        Xs, its_s, rr_s = eng.solve(SOLVER_HID, 0, B, 1e-12, 200)
        nrm = np.linalg.norm(B.T, axis=0)
        true = np.linalg.norm(B.T - A @ Xs.T, axis=0) / np.where(nrm > 0, nrm, 1.0)
        assert true.max() < 1.5e-13
        assert np.max(np.abs(np.asarray(its_s) - out[1][1])) <= 1
        eng.set_option("stop_factor", 1.0)
        # iteration budget too small: reported, not raised; the iterate is the best found so far
        Xn, its_n, rr_n = eng.solve(SOLVER_HID, 0, B[:8], 1e-12, 4)
        assert rr_n[0] > 1e-12 and np.isfinite(Xn).all()
        lu = rp.LUSolver(A)
        assert _relerr(out[1][0][0], lu(B[0])) < 1e-9
    finally:
        eng.set_option("gram_cycle", 1)
        eng.set_option("stop_factor", 1.0)
    eng.close()


def test_small_reference_levels_solved_directly_match_the_iterative_solves():
    """sw_setup_level_inverse / option direct_small: the reference hierarchy's small intermediate levels
    (2048 rows on schwinger128) carry the dense inverse of their operator, formed on the device; solves that
    start there -- the MLMC coarse solve A_2^-1 R_1 R_0 x (utils.py:306-329) and the level-2 difference
    level's fine solve -- are x = A^-1 b, x += A^-1 (b - A x) instead of a multigrid-preconditioned FGMRES.
    Same solutions to the solver tolerance, true residual below it; MLMC estimates unchanged to 1e-10."""
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc")
    tp['mlmc_deflat_vctrs'] = [0] * 3
    mg = MG(A)
    mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
             acc_eigvs=tp['accuracy_mg_eigvs'], sys_type='schwinger', params=tp)
    eng = mg.engine
    levels = mg.ml.levels
    assert [lev.A.shape[0] for lev in levels] == [32768, 8192, 2048, 512]
    A2 = levels[2].A
    B = _rand((70, 2048), 3)
    out = {}
    try:
        for flag in (1, 0):
            eng.set_option("direct_small", flag)
            X, its, rr = eng.solve(REF_HID, 2, B, 1e-12, 2000)
            true = np.linalg.norm(B.T - A2 @ X.T, axis=0) / np.linalg.norm(B.T, axis=0)
            assert true.max() < 2e-12, (flag, true.max())
            out[flag] = (X, np.asarray(its))
            np.random.seed(31)
            probes = utils.draw_probes(8, A.shape[0])
            e_skip, _, itc = eng.hutch_batch(MODE_MLMC_SKIP, 0, probes, 1e-12, 1000)
            np.random.seed(32)
            probes2 = utils.draw_probes(8, 2048)
            e_l2, itf2, _ = eng.hutch_batch(MODE_MLMC, 2, probes2, 1e-12, 1000)
            out[(flag, "e")] = (e_skip, e_l2, np.asarray(itc), np.asarray(itf2))
    finally:
        eng.set_option("direct_small", 1)
    assert np.all(out[1][1] == 1) and out[0][1].max() > 1               # direct: one "iteration"; iterative: many
    assert _relerr(out[1][0], out[0][0]) < 1e-10
    assert np.all(out[(1, "e")][2] == 1) and np.all(out[(1, "e")][3] == 1)
    scale = np.maximum(np.abs(out[(0, "e")][0]), 1.0)
    assert np.max(np.abs(out[(1, "e")][0] - out[(0, "e")][0]) / scale) < 1e-9
    assert np.max(np.abs(out[(1, "e")][1] - out[(0, "e")][1]) / np.maximum(np.abs(out[(0, "e")][1]), 1.0)) < 1e-9
    eng.close()
