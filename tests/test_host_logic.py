"""CPU: host-side logic of the product package (no GPU): hierarchy construction against the
oracle's literal restatement, presets, parameter plumbing, error behaviour, the batched probe
loop against the sequential stopping rule, and the probe stream."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from deflatedmlmc_schwinger_amd import gateway, hierarchy, matrix, stoch_trace, utils
from deflatedmlmc_schwinger_amd.engine import EngineError, ProbeStream, device_count
from oracle import ref_path as rp

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="module")
def A16():
    p = gateway.set_params('schwinger16')
    return matrix.loadMatrix(p['matrix'], p['matrix_params'])


@pytest.fixture(scope="module")
def A128():
    p = gateway.set_params('schwinger128')
    return matrix.loadMatrix(p['matrix'], p['matrix_params'])


def test_links_rebuild_matrix_bit_exact(A16, A128):
    for A, L in ((A16, 16), (A128, 128)):
        lat = hierarchy.detect_lattice(A)
        assert lat is not None and lat[0] == L
        _, mass, U1, U2 = lat
        assert abs(np.abs(U1) - 1).max() < 1e-15
        S = hierarchy.wilson_from_links(U1, U2, L)
        d = abs(S + mass * sp.identity(A.shape[0]) - A)
        assert (d.max() if d.nnz else 0.0) == 0.0
        # same operator as the oracle's restatement of the stencil
        d2 = abs(S - rp.build_wilson(U1, U2, L))
        assert (d2.max() if d2.nnz else 0.0) == 0.0
    # row 0 of the 128^2 matrix
    assert sorted(A128[0].indices.tolist()) == [0, 1, 127, 128, 16256, 16385, 16511, 16512, 32640]
    assert hierarchy.detect_lattice(sp.identity(512, format="csr") * 2.0) is None


def test_shift_operator_matches_reference_construction():
    for n, s in ((512, 32), (32768, 512), (64, 8)):
        mine = hierarchy.shift_operator(n, s)
        ref = rp.pperm_matrix(n, s)
        assert abs(mine - ref).nnz == 0
        v = np.arange(n, dtype=float)
        assert np.array_equal(mine.transpose() @ v, np.roll(v, s))       # (P^T v)[i] = v[i-s]


@pytest.mark.parametrize("name", ["schwinger16", "schwinger128"])
def test_reference_hierarchy_matches_oracle_restatement(name, A16, A128):
    """vectorised builder vs the literal loops of multigrid.py:192-259 on the same test vectors."""
    A = A16 if name == "schwinger16" else A128
    p = gateway.set_params(name)
    p['function_tol'] = 1e-12
    tp = utils.trace_params_from_params(p, "mlmc")
    if name == "schwinger16":
        tp['use_permuted'], tp['x_displacement'] = True, 1       # exercise the Pperm bookkeeping
    ml, cinv, tv = hierarchy.reference_hierarchy(A, tp['dof'], tp['aggrs'], tp['max_nr_levels'],
                                                 tp['accuracy_mg_eigvs'], tp)
    lev, cinv_o, _ = rp.mg_setup(A, tp['dof'], tp['aggrs'], tp['max_nr_levels'],
                                 tp['accuracy_mg_eigvs'], tp, testvectors=tv)
    assert len(ml.levels) == len(lev) == tp['max_nr_levels']
    for a, b in zip(ml.levels, lev):
        assert a.A.shape == b.A.shape
        assert abs(a.A - b.A).max() < 1e-9
        assert a.perm_shift == b.perm_shift
        if not isinstance(b.P, int):
            # single-pass Gram-Schmidt on nearly dependent local vectors amplifies rounding
            # (||P^H P - I|| is 1.7e-12, not 1e-15), hence not 1e-15 here either
            assert abs(a.P - b.P).max() < 1e-9
            assert a.P.nnz == b.P.nnz
            # check_quality_MG diagnostics (multigrid.py:282-316) as assertions
            PhP = (a.R @ a.P).toarray()
            assert abs(PhP - np.eye(PhP.shape[0])).max() < 1e-10
        assert abs(a.Bblock_perm - b.Bblock_perm).max() < 1e-9
        n = a.A.shape[0]
        g3A = sp.diags(np.where(np.arange(n) < n // 2, 1.0, -1.0)) @ a.A
        assert abs(g3A - g3A.getH()).max() < 1e-12          # gamma3-hermiticity on every level
    assert np.abs(np.asarray(cinv) - cinv_o).max() / np.abs(cinv_o).max() < 1e-7
    if name == "schwinger128":
        assert [l.A.shape[0] for l in ml.levels] == [32768, 8192, 2048, 512]
        assert [l.perm_shift for l in ml.levels] == [512, 128, 32, 8]
        assert [ml.levels[i].P.nnz for i in range(3)] == [131072, 32768, 8192]


def test_solver_hierarchy_properties(A16):
    cfg = dict(hierarchy.DEFAULT_SOLVER_CFG, coarsening=[(4, 4), (2, 4)])
    sh = hierarchy.solver_hierarchy(A16, 16, cfg)
    assert [a.shape[0] for a in sh["A"]] == [512, 128, 32]
    for P in sh["P"]:
        PhP = (P.conj().T @ P).toarray()
        assert abs(PhP - np.eye(PhP.shape[0])).max() < 1e-12
    # Galerkin and chirality preservation
    A1 = (sh["P"][0].conj().T @ A16 @ sh["P"][0]).toarray()
    assert abs(A1 - sh["A"][1].toarray()).max() < 1e-13
    w = hierarchy.smoother_weights(A16, 4)
    assert w.shape == (4,) and np.all(np.isfinite(w))
    # the polynomial prod(1 - w_k z) is a residual polynomial: it damps a random vector
    rng = np.random.default_rng(0)
    b = rng.standard_normal(512) + 1j * rng.standard_normal(512)
    r = b.copy()
    for wk in w:
        r = r - wk * (A16 @ r)
    assert np.linalg.norm(r) < np.linalg.norm(b)


def test_presets_and_param_plumbing():
    p = gateway.set_params('schwinger128')
    assert p['matrix'] == 'schwinger128.mat' and p['matrix_params']['mass'] == -0.1320
    assert p['aggrs'] == [16, 4, 4] and p['dof'] == [2, 8, 8, 8] and p['max_nr_levels'] == 4
    assert p['mlmc_levels_to_skip'] == [1] and p['nr_deflat_vctrs'] == 8
    assert p['use_permuted'] is True and p['x_displacement'] == 2 and p['latt_dims'] == [128, 128]
    assert np.random.get_state()[1][0] == np.random.RandomState(51234).get_state()[1][0]
    with pytest.raises(Exception, match="Non-existent option"):
        gateway.set_params('schwinger64')
    p['function_tol'] = 1e-12
    for kind in ("mlmc", "hutchinson"):
        mine = utils.trace_params_from_params(p, kind)
        gold = G["trace_params_" + kind]
        # the golden dict came from the reference's utils.trace_params_from_params
        assert mine == gold
    with pytest.raises(Exception, match="not available"):
        utils.trace_params_from_params(p, "other")
    with pytest.raises(Exception, match="not available"):
        utils.print_post_results(None, p, {}, "other")
    p16 = gateway.set_params('schwinger16')
    p16['function_tol'] = 1e-12
    utils.trace_params_from_params(p16, "mlmc")           # the completed preset has every key


def test_flops_model_matches_reference_utils():
    class L:
        def __init__(self, nnz):
            self.A = type("M", (), {"nnz": nnz})()

    class S:
        smooth_iters = 2
    lv = [L(294912), L(294904), L(98304), L(24576)]
    for key, val in G["flopsV_manual"].items():
        a, b = eval(key)
        assert utils.flopsV_manual(a, lv, b, S()) == val


def test_custom_timer_semantics():
    t = utils.CustomTimer()
    t.start("mvm")
    with pytest.raises(Exception, match="already timing"):
        t.start("mvm")
    t.end("mvm")
    with pytest.raises(Exception, match="already down"):
        t.end("mvm")
    t.start("axpy")
    with pytest.raises(Exception, match="Uknown part"):
        t.end("nonsense")
    assert "matrix-vector multiplications" in str(t)
    t.reset()
    assert t.mvm == 0.0


def test_probe_stream_matches_numpy_global_stream(lib_built):
    np.random.seed(123456)
    ref = (2 * np.random.randint(2, size=(3, 1000)) - 1).astype(np.int8)
    ps = ProbeStream(123456)
    assert ps.raw(4).tolist() == G["mt19937_seed123456_first_words"][:4]
    ps = ProbeStream(123456)
    assert np.array_equal(ps.rademacher(3, 1000), ref)
    # skip == draw-and-discard; draw_probes == np.random.randint calls of the reference
    ps = ProbeStream(123456)
    ps.skip(2000)
    assert np.array_equal(ps.rademacher(1, 1000)[0], ref[2])
    np.random.seed(123456)
    assert np.array_equal(utils.draw_probes(3, 1000), ref)
    ps = ProbeStream(51234)
    np.random.seed(51234)
    assert np.array_equal(ps.rademacher(2, 700), utils.draw_probes(2, 700))


def test_run_probe_loop_replays_sequential_rule():
    """batched rounds + replay == the one-by-one loop of stoch_trace.py:137-154, including the
    position of the global NumPy stream on exit."""
    n = 64
    rng = np.random.default_rng(3)
    w = rng.standard_normal(n) + 1j * rng.standard_normal(n)

    def evaluate(probes):
        p = probes.astype(np.complex128)
        e = (p * w).sum(axis=1) * 3.0
        return e, np.full(len(e), 7), np.zeros(len(e), dtype=np.int64)

    tol = 1.3
    np.random.seed(99)
    seq = []
    while True:
        x = rp.rademacher(n)
        seq.append(((x * w).sum() * 3.0))
        idx, avg, dev = rp.stopping_rule(np.array(seq), tol)
        if idx == len(seq) - 1 and len(seq) > 5 and dev / np.sqrt(len(seq)) < tol:
            break
        if len(seq) > 5000:
            break
    after_seq = np.random.randint(1 << 30)
    for batch in (1, 7, 64, 256):
        np.random.seed(99)
        out = stoch_trace.run_probe_loop(evaluate, n, tol, 100000, batch)
        assert out["index"] == idx
        assert out["avg"] == avg and out["dev"] == dev
        assert np.array_equal(out["ests"], np.array(seq))
        assert int(out["iters_fine"].sum()) == 7 * (idx + 1)
        assert np.random.randint(1 << 30) == after_seq
    # max_nr_ests cap
    np.random.seed(99)
    out = stoch_trace.run_probe_loop(evaluate, n, 0.0, 10, 4)
    assert out["index"] == 9 and len(out["ests"]) == 10


def test_estimator_option_validation(A16):
    p = gateway.set_params('schwinger16')
    p['function_tol'] = 1e-12
    tp = utils.trace_params_from_params(p, "mlmc")
    tp['mlmc_levels_to_skip'] = [1, 2]
    with pytest.raises(Exception, match="Only allowed to skip one level"):
        stoch_trace.mlmc(A16, tp)
    tp['mlmc_levels_to_skip'] = [2]
    with pytest.raises(Exception, match="skip the second level"):
        stoch_trace.mlmc(A16, tp)
    bad = dict(tp, test_vectors_type="nonsense")
    with pytest.raises(Exception, match="unknown type of test vectors"):
        hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], 3, 'low', bad)
    with pytest.raises(Exception, match="accuracy_mg_eigvs"):
        hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], 3, 'medium', tp)


def test_reference_hierarchy_eigensolver_hook_and_deferred_inverse(A16):
    """hierarchy.reference_hierarchy's hooks for the device setup: `eigs_fn(level, A_l, nvec, tol)` supplies the
    test vectors of the levels it answers for (None = the reference's own ARPACK call), a partial `testvectors`
    list pins the levels it names, `invert=False` leaves the coarsest inverse to the engine -- the hierarchy built
    through the hooks from the SAME vectors equals the plain one entry for entry."""
    p = gateway.set_params('schwinger16')
    p['function_tol'] = 1e-12
    tp = utils.trace_params_from_params(p, "mlmc")
    ml, cinv, tv = hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], tp['max_nr_levels'],
                                                 tp['accuracy_mg_eigvs'], tp)
    calls = []

    def eigs_fn(level, Al, nvec, tol):
        calls.append((level, Al.shape[0], nvec, tol))
        return tv[level] if level == 0 else None          # level 0 from the "device", the rest from ARPACK

    ml2, cinv2, tv2 = hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], tp['max_nr_levels'],
                                                    tp['accuracy_mg_eigvs'], tp, eigs_fn=eigs_fn, invert=False)
    assert cinv2 is None and cinv is not None
    assert [c[0] for c in calls] == list(range(tp['max_nr_levels'] - 1))
    assert calls[0][1:] == (A16.shape[0], int(tp['dof'][1] / 2), 1.0e-3)
    assert np.array_equal(np.asarray(tv2[0]), np.asarray(tv[0]))
    assert abs(ml2.levels[1].A - ml.levels[1].A).max() == 0.0          # same vectors, same arithmetic
    # a partial list of test vectors pins level 0 only
    ml3, _, tv3 = hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], tp['max_nr_levels'],
                                                tp['accuracy_mg_eigvs'], tp, testvectors=[tv[0]])
    assert len(tv3) == tp['max_nr_levels'] - 1 and abs(ml3.levels[1].A - ml.levels[1].A).max() == 0.0
    # the lazy attribute: without an engine the MG object hands back what was set
    from deflatedmlmc_schwinger_amd.multigrid import MG
    mg = MG(A16)
    assert mg.coarsest_inv == []
    mg.coarsest_inv = cinv
    assert mg.coarsest_inv is cinv
    mg.coarsest_inv = None
    assert mg.coarsest_inv is None


def test_product_fails_loudly_without_gpu(A16):
    """no CPU fallback: without a HIP device the product path raises."""
    if device_count() > 0:
        pytest.skip("a GPU is present")
    from deflatedmlmc_schwinger_amd.multigrid import MG
    p = gateway.set_params('schwinger16')
    p['function_tol'] = 1e-12
    tp = utils.trace_params_from_params(p, "hutchinson")
    mg = MG(A16)
    with pytest.raises(EngineError, match="no HIP device"):
        mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=3, dim=2, acc_eigvs='low',
                 sys_type='schwinger', params=tp)
    with pytest.raises(EngineError):
        MG(A16).solve(A16, np.ones(512), 1e-12)


def test_setup_cache_roundtrip_and_run_report(tmp_path, A16):
    from deflatedmlmc_schwinger_amd import cache
    key = cache.matrix_key(A16, {"k": 8})
    assert key == cache.matrix_key(A16, {"k": 8}) and key != cache.matrix_key(A16, {"k": 9})
    assert cache.load(str(tmp_path), "defl", key) is None
    cache.save(str(tmp_path), "defl", key, {"S": np.arange(3.0), "V": np.eye(3, dtype=complex)})
    hit = cache.load(str(tmp_path), "defl", key)
    assert np.array_equal(hit["S"], np.arange(3.0)) and hit["V"].dtype == np.complex128
    # the reference hierarchy rebuilt from cached test vectors equals the first build
    p = gateway.set_params('schwinger16')
    p['function_tol'] = 1e-12
    tp = utils.trace_params_from_params(p, "mlmc")
    ml1, _, tv = hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], 3, 'low', tp)
    ml2, _, _ = hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], 3, 'low', tp, testvectors=tv)
    assert abs(ml1.levels[2].A - ml2.levels[2].A).max() == 0.0
    rep = tmp_path / "runs.jsonl"
    rec = cache.write_run_report({"trace": 1 + 2j, "std_dev": 3.0, "nr_ests": 9, "function_iters": 50},
                                 "hutchinson", {"report_path": str(rep), "matrix": "x.mat"}, 2.0)
    assert rec["probe_samples_per_s"] == 5.0
    assert json.loads(rep.read_text().splitlines()[0])["trace"] == [1.0, 2.0]


@pytest.mark.parametrize("tv_type", ["LSVs", "RSVs"])
def test_singular_vector_test_vectors_give_a_g3_compatible_hierarchy(A16, tv_type):
    """test_vectors_type LSVs / RSVs (multigrid.py:159-188): eigenvectors of Q = g3 A; the checks
    are the reference's own check_quality_MG norms (multigrid.py:282-316)."""
    p = gateway.set_params('schwinger16')
    p['function_tol'] = 1e-12
    p['test_vectors_type'] = tv_type
    p['accuracy_mg_eigvs'] = 'high'
    tp = utils.trace_params_from_params(p, "mlmc")
    ml, cinv, tvs = hierarchy.reference_hierarchy(A16, tp['dof'], tp['aggrs'], tp['max_nr_levels'],
                                                  'high', tp)
    assert len(ml.levels) == 3
    for i in range(2):
        lev, nxt = ml.levels[i], ml.levels[i + 1]
        P = lev.P
        nc = P.shape[1]
        g3c = sp.diags([np.concatenate([np.ones(nc // 2), -np.ones(nc // 2)])], [0])
        assert abs(lev.g3 @ P - P @ g3c).max() < 1e-13            # gamma3-compatibility
        PhP = (P.conjugate().transpose() @ P).toarray()
        assert np.abs(PhP - np.eye(nc)).max() < 1e-10             # single CGS sweep: 1e-11-ish
        Qc = (g3c @ nxt.A).toarray()
        assert np.abs(Qc - Qc.conj().T).max() < 1e-12             # g3 A_c hermitian
    # the singular vectors themselves: Q v = lambda v  with |lambda| the smallest singular values
    n = A16.shape[0]
    g3 = ml.levels[0].g3
    v = tvs[0][:, 0].copy()
    if tv_type == "LSVs":
        v[n // 2:] = -v[n // 2:]                                   # back to the eigenvector of Q
    Qv = g3 @ (A16 @ v)
    lam = np.vdot(v, Qv) / np.vdot(v, v)
    assert np.linalg.norm(Qv - lam * v) < 1e-7 * np.linalg.norm(v)
    smin = np.linalg.svd(A16.toarray(), compute_uv=False)[-1]
    assert abs(abs(lam) - smin) < 1e-6 * smin + 1e-9 or abs(lam) < 10 * smin


@pytest.mark.parametrize("Lf,hd,agg,fine", [(16, 1, 4, True), (32, 1, 8, True), (8, 8, 2, False),
                                            (16, 8, 4, False)])
def test_device_setup_geometry_reproduces_the_host_prolongator(Lf, hd, agg, fine):
    """setup_gpu.level_geometry (block membership + grouped-ELL structure handed to
    sw_setup_transfer) against hierarchy._site_prolongator: with the same per-block Q the
    prolongator is identical entry for entry.  Lattice level: groups of 8 fine rows where four sites of
    one parity along x stay inside an aggregate (edge 8), groups of 4 otherwise (edge 4)."""
    from deflatedmlmc_schwinger_amd import setup_gpu
    g = setup_gpu.level_geometry(Lf, hd, agg, fine)
    assert g["G"] == ((8 if agg % 8 == 0 else 4) if fine else 8)
    n = 2 * Lf * Lf * hd
    rng = np.random.default_rng(1)
    tv = rng.standard_normal((n, 8)) + 1j * rng.standard_normal((n, 8))
    P = hierarchy._site_prolongator(sp.identity(n, dtype=np.complex128, format='csr'), Lf, hd, agg, 8,
                                    tv, fine)
    idx = np.arange(n)
    if fine:
        V = Lf * Lf
        half, site = idx // V, idx % V
        x, y = site % Lf, site // Lf
        internal = (((x + y) & 1) * (V // 2) + (site >> 1)) * 2 + half
    else:
        internal = idx
    tvi = np.empty_like(tv)
    tvi[internal] = tv
    nblocks, rpb = g["blk_rows"].shape
    assert sorted(g["blk_rows"].reshape(-1).tolist()) == list(range(n))      # a partition of the rows
    Q = np.stack([np.linalg.qr(tvi[g["blk_rows"][b]])[0] for b in range(nblocks)]).reshape(-1)
    vals = np.where(g["pmap"] >= 0, Q[np.clip(g["pmap"], 0, None)], 0)
    xc = rng.standard_normal(g["n_c"]) + 1j * rng.standard_normal(g["n_c"])
    y_dev = np.einsum('gkr,gk->gr', vals, xc[g["pcols"]]).reshape(-1)
    y_ref = np.empty(n, complex)
    y_ref[internal] = P @ xc
    assert np.abs(y_dev - y_ref).max() == 0.0
    # neighbour lists: five distinct sites with distinct probing colours
    Lc = g["Lc"]
    nbr = g["nbr"]
    col = (nbr % Lc) % 4 + 4 * ((nbr // Lc) % 4)
    assert all(len(set(r)) == 5 for r in nbr.tolist()) and all(len(set(r)) == 5 for r in col.tolist())


def test_block_row_packers_put_the_own_site_block_last():
    """The smoother kernel reads its own X rows from the operand registers of the last four k-steps
    (k_bsr_mfma, xreg): hierarchy.block_rows_from_matrix and setup_gpu.level_geometry must put the
    block / neighbour of the row's own site last, and the packed form must still be the matrix."""
    import scipy.sparse as sp
    from deflatedmlmc_schwinger_amd import hierarchy, setup_gpu
    rng = np.random.default_rng(5)
    Lc = 4
    ns, n = Lc * Lc, Lc * Lc * 16
    site = np.arange(ns)
    x, y = site % Lc, site // Lc
    blocks = {}
    for s_ in site:
        for t in {int(s_), int(y[s_] * Lc + (x[s_] + 1) % Lc), int(y[s_] * Lc + (x[s_] - 1) % Lc),
                  int(((y[s_] + 1) % Lc) * Lc + x[s_]), int(((y[s_] - 1) % Lc) * Lc + x[s_])}:
            blocks[(int(s_), t)] = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
    A = sp.bmat([[blocks.get((i, j)) for j in range(ns)] for i in range(ns)], format="csr")
    rows = np.arange(0, ns, 2)
    tmap, kcol, vals = hierarchy.block_rows_from_matrix(A, rows, n)
    KS = kcol.shape[1]
    assert KS % 4 == 0
    for r, s_ in enumerate(rows):
        assert list(kcol[r, KS - 4:]) == [16 * s_ + 4 * g for g in range(4)]
    xv = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = A @ xv
    for r, s_ in enumerate(rows):
        acc = np.zeros(16, dtype=complex)
        for k in range(KS):
            for c in range(4):
                acc += vals[r, k, 16 * c:16 * c + 16] * xv[kcol[r, k] + c]
        assert abs(acc - ref[16 * s_:16 * s_ + 16]).max() < 1e-12
        assert tmap[r] == s_
    g = setup_gpu.level_geometry(32, 1, 4, True)
    nbr = g["nbr"]
    assert (nbr[:, 4] == np.arange(nbr.shape[0])).all()
    assert all(len(set(row)) == 5 for row in nbr)


def test_f32_capable_configurations():
    from deflatedmlmc_schwinger_amd import hierarchy
    assert hierarchy.f32_capable(hierarchy.TUNED_SOLVER_CFG_128)
    assert not hierarchy.f32_capable(hierarchy.DEFAULT_SOLVER_CFG)             # level 0 not even-odd
    assert hierarchy.f32_capable(dict(hierarchy.TUNED_SOLVER_CFG_128, cycle=[(0, 6, 2), (0, 9, 0)]))
    assert not hierarchy.f32_capable(dict(hierarchy.TUNED_SOLVER_CFG_128, cycle=[(2, 6, 0), (0, 9, 0)]))
    assert not hierarchy.f32_capable(dict(hierarchy.TUNED_SOLVER_CFG_128, smoother="mr"))


def test_batched_even_odd_schur_construction_equals_the_sparse_one():
    """hierarchy.coarse_schur_blocks (batched 16 x 16 algebra straight into block-row form) against
    hierarchy.coarse_schur_operators (sparse products) on a random 5-point block operator, through the
    packed form: S (own block last), F, G, Hb and the compressed S_ee used for the smoother polynomial;
    plus the round trip between site blocks and the engine's block-row layout."""
    import scipy.sparse as sp
    from deflatedmlmc_schwinger_amd import hierarchy as H
    rng = np.random.default_rng(3)
    Lc = 8
    ns, n = Lc * Lc, Lc * Lc * 16
    site = np.arange(ns)
    x, y = site % Lc, site // Lc
    blocks = {}
    for s_ in site:
        for t in [int(s_), int(y[s_] * Lc + (x[s_] + 1) % Lc), int(y[s_] * Lc + (x[s_] - 1) % Lc),
                  int(((y[s_] + 1) % Lc) * Lc + x[s_]), int(((y[s_] - 1) % Lc) * Lc + x[s_])]:
            b = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
            blocks[(int(s_), t)] = b + (8 * np.eye(16) if t == s_ else 0)
    A = sp.bmat([[blocks.get((i, j)) for j in range(ns)] for i in range(ns)], format="csr")
    old = H.coarse_schur_operators(A, Lc)
    nbr, blk = H.site_blocks_of(A, Lc)
    new = H.coarse_schur_blocks(nbr, blk, Lc)
    assert new is not None

    def unpack(tmap, kcol, vals):
        RT, KS = kcol.shape
        r = np.repeat(tmap.astype(np.int64) * 16, KS * 64).reshape(RT, KS, 4, 16) + np.arange(16)
        c = kcol.astype(np.int64)[:, :, None, None] + np.arange(4)[None, None, :, None] + np.zeros((1, 1, 1, 16), int)
        v = vals.reshape(RT, KS, 4, 16)
        return sp.csr_matrix((v.ravel(), (r.ravel(), c.ravel())), shape=(n, n))

    for name, pk in zip(("S", "F", "G", "Hb"), new["packed"]):
        assert abs(unpack(*pk) - old[name]).max() < 1e-12, name
    E = old["E_rows"]
    assert abs(new["S_ee"] - old["S"][E][:, E]).max() < 1e-12
    assert (new["E_rows"] == old["E_rows"]).all() and (new["O_rows"] == old["O_rows"]).all()
    tm, kc, _ = new["packed"][0]
    assert all(list(kc[r, -4:]) == [16 * tm[r] + 4 * g for g in range(4)] for r in range(len(tm)))
    kcA, vlA = H.pack_site_blocks(blk, nbr)
    nbr2, blk2 = H.site_blocks_from_block_rows(kcA, vlA)
    assert (nbr2 == nbr).all() and abs(blk2 - blk).max() == 0.0


def test_even_odd_reduced_system_algebra_behind_the_outer_solver(A16):
    """What the engine's fgmres_eo (outer solve on half vectors) relies on, on 16^2 in NumPy:
    (i) with x_o set from x_e the full residual has a vanishing odd half and its even half is the
    residual of the reduced system; (ii) the even block of the even-odd smoothed cycle applied to
    (r_e; 0) needs neither the hop before the Schur steps nor the odd half of the restriction's input;
    (iii) the reduced solve reaches the LU solution with the full system's stopping criterion in as many
    iterations (+-1) as the full-system FGMRES with the same cycle."""
    from oracle import engine_model as em
    L = 16
    cfg = dict(hierarchy.DEFAULT_SOLVER_CFG, coarsening=[(4, 4), (2, 4)], cycle=[(0, 4, 0), (0, 5, 0)])
    sh = hierarchy.solver_hierarchy(A16, L, cfg)
    A = sp.csr_matrix(A16)
    S, E, O, D = hierarchy.schur_complement(A, L)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    Aeo, Aoe = A[E][:, O], A[O][:, E]
    # (i)
    xe = rng.standard_normal((E.size, 3)) + 1j * rng.standard_normal((E.size, 3))
    X = np.zeros_like(B)
    X[E] = xe
    X[O] = (B[O] - Aoe @ xe) / D
    R = B - A @ X
    bp = B[E] - Aeo @ (B[O] / D)
    assert np.abs(R[O]).max() < 1e-13
    assert np.abs(R[E] - (bp - S @ xe)).max() < 1e-12
    # (ii)
    w1 = hierarchy.smoother_weights(sh["A"][1], 5)
    w_eo = hierarchy.smoother_weights(S, 4)
    weights = [None, (np.zeros(0), w1)]
    cyc = [tuple(c) for c in cfg["cycle"]]
    cinv = hierarchy.dense_inverse(sh["A"][2].toarray())
    M = lambda V: em.cycle_eo(sh["A"], sh["P"], cinv, cyc, V, weights, w_eo, E, O, D)   # noqa: E731
    re = rng.standard_normal((E.size, 3)) + 1j * rng.standard_normal((E.size, 3))
    full = np.zeros((n, 3), dtype=complex)
    full[E] = re
    ref = M(full)[E]
    P0 = sh["P"][0]
    Re = sp.csr_matrix(P0.conj().T)[:, E]                       # restrictor, even columns only
    xc = em.cycle(sh["A"], sh["P"], cinv, cyc, 1, Re @ re, weights)
    ze = (P0 @ xc)[E]
    for w in w_eo:
        ze = ze + w * (re - S @ ze)                             # b' = r_e itself: no hop
    assert np.linalg.norm(ze - ref) / np.linalg.norm(ref) < 1e-12
    # (iii)
    Xr, its_r = em.solve_even_odd_reduced(A, B, M, E, O, D, 1e-12, 3, 60)
    Xf, its_f = em.fgmres_restarted(lambda V: A @ V, B, M, 1e-12, 3, 60)
    lu = rp.LUSolver(A16)
    ref = np.stack([lu(B[:, k]) for k in range(3)], axis=1)
    nb_ = np.linalg.norm(B, axis=0)
    assert (np.linalg.norm(B - A @ Xr, axis=0) / nb_).max() < 1e-12
    assert np.linalg.norm(Xr - ref) / np.linalg.norm(ref) < 1e-10
    assert np.linalg.norm(Xf - ref) / np.linalg.norm(ref) < 1e-10
    assert abs(its_r - its_f) <= 3, (its_r, its_f)      # restart granularity of the lock-step model


def test_dense_schur_inverse_in_block_row_form_inverts_the_schur_operator():
    """hierarchy.dense_schur_inverse_blocks (even-odd operator 4 of a block level: the level solved exactly
    instead of smoothed) against the Schur operator of coarse_schur_blocks, both read back from their packed
    MFMA block-row forms the way the engine indexes them: S^-1 S = I on the even rows, nothing elsewhere."""
    Lc = 8
    ns = Lc * Lc
    rng = np.random.default_rng(1)
    site = np.arange(ns)
    xs, ys = site % Lc, site // Lc
    nbr = np.stack([ys * Lc + (xs + 1) % Lc, ys * Lc + (xs - 1) % Lc, ((ys + 1) % Lc) * Lc + xs,
                    ((ys - 1) % Lc) * Lc + xs, site], axis=1)
    blk = 0.1 * (rng.standard_normal((ns, 5, 16, 16)) + 1j * rng.standard_normal((ns, 5, 16, 16)))
    blk[:, 4] += 4 * np.eye(16)
    ops = hierarchy.coarse_schur_blocks(nbr, blk, Lc)
    tmap, kcol, vals = hierarchy.dense_schur_inverse_blocks(ops)
    n = ns * 16

    def as_matrix(tmap, kcol, vals):
        RT, KS = kcol.shape
        r = np.repeat(tmap.astype(np.int64) * 16, KS * 64).reshape(RT, KS, 4, 16) + np.arange(16)
        c = kcol.astype(np.int64)[:, :, None, None] + np.arange(4)[None, None, :, None] + np.zeros((1, 1, 1, 16), int)
        return sp.csr_matrix((vals.reshape(RT, KS, 4, 16).ravel(), (r.ravel(), c.ravel())), shape=(n, n))

    prod = (as_matrix(tmap, kcol, vals) @ as_matrix(*ops["packed"][0])).toarray()
    Er = ops["E_rows"]
    assert abs(prod[np.ix_(Er, Er)] - np.eye(Er.size)).max() < 1e-12
    prod[np.ix_(Er, Er)] = 0
    assert abs(prod).max() == 0
    assert kcol.shape == (ns // 2, 4 * (ns // 2)) and vals.shape == (ns // 2, 4 * (ns // 2), 64)


def test_screened_stopping_rule_equals_the_reference_replay():
    """stoch_trace.first_stop_index (running-sum screen + exact re-evaluation of the candidates)
    against the reference's own O(N^2) loop (oracle restatement of stoch_trace.py:137-154): identical
    stop index, mean and deviation -- on 10^5 synthetic estimates with the statistics of the 128^2
    deflated estimator (mean ~ -7+6j, deviation ~ 140), on borderline tolerances, on sequences with a
    large offset (where a one-pass variance would cancel) and on rounds that resume mid-sequence."""
    from deflatedmlmc_schwinger_amd import stoch_trace
    from oracle import ref_path as rp
    rng = np.random.default_rng(99)
    big = (-7.0 + 6.0j) + 100.0 * (rng.standard_normal(100000) + 1j * rng.standard_normal(100000))
    # a tolerance reached about a third of the way in: 141 / sqrt(i + 1) < tol  =>  i ~ 3.2e4
    tol = 0.79
    i_ref, avg_ref, dev_ref = rp.stopping_rule(big[:40000], tol)
    assert 20000 < i_ref < 39999
    hit = stoch_trace.first_stop_index(big, 0, tol)
    assert hit is not None and hit[0] == i_ref and hit[1] == avg_ref and hit[2] == dev_ref
    # resumed rounds: scanning from first_new on finds the same index; beyond it, the next one
    assert stoch_trace.first_stop_index(big, i_ref - 100, tol)[0] == i_ref
    assert stoch_trace.first_stop_index(big[:i_ref], 0, tol) is None
    # never reached
    assert stoch_trace.first_stop_index(big, 0, 1e-3) is None
    # many short sequences, tolerances placed exactly ON the error estimate of some index (borderline)
    for trial in range(200):
        n = int(rng.integers(6, 400))
        off = (10.0 ** rng.integers(0, 9)) * (1.0 + 0.5j) if trial % 3 == 0 else 0.0
        e = off + rng.standard_normal(n) * (1.0 + rng.random()) + 1j * rng.standard_normal(n)
        k = int(rng.integers(5, n))
        cur = e[:k + 1]
        dev = np.sqrt(np.sum(np.abs(cur - np.sum(cur) / (k + 1)) ** 2) / (k + 1))
        for tol_t in (dev / np.sqrt(k + 1), np.nextafter(dev / np.sqrt(k + 1), np.inf), 0.3, 0.05):
            i_ref, avg_ref, dev_ref = rp.stopping_rule(e, tol_t)
            stopped = i_ref < n - 1 or (dev_ref / np.sqrt(n) < tol_t and n - 1 >= 5)
            hit = stoch_trace.first_stop_index(e, 0, tol_t)
            if stopped:
                assert hit is not None and hit[0] == i_ref and hit[1] == avg_ref and hit[2] == dev_ref, trial
            else:
                assert hit is None, trial


def test_reference_coarse_level_in_even_odd_form_solves_the_same_system(A128):
    """hierarchy.reference_coarse_eo (build key ref_coarsest = "eo", BASELINE config 2 as written): the coarse
    level of the reference's 2-level aggregation (strips of 32 sites along x at fixed spin and y, 8 coarse dofs
    each) reordered tile by tile -- both spin strips of one (y, x-strip) = 16 rows -- is a nearest-neighbour
    stencil of 16 x 16 blocks that two colours split; x_e = S^-1 (b_e - F b_o), x_o = G b_o - Hb x_e with the
    packed operators must equal the sparse-LU solve of the permuted coarse operator; pi is a permutation that
    keeps the 8 dofs of an aggregate together; layouts it does not fit are refused."""
    import scipy.sparse.linalg as spla
    n = A128.shape[0]
    rng = np.random.default_rng(0)
    tv = rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4))
    P = hierarchy.prolongator_from_testvectors(tv, n, 0, [2, 8, 8], [16, 4])
    Ac = sp.csr_matrix(P.conj().T @ A128 @ P)
    res = hierarchy.reference_coarse_eo(P, Ac, 128, 32)
    assert res is not None
    pi, packed = res
    nc = Ac.shape[0]
    assert np.array_equal(np.sort(pi), np.arange(nc))
    assert np.all(np.diff(pi.reshape(-1, 8), axis=1) == 1) and np.all(pi.reshape(-1, 8)[:, 0] % 8 == 0)
    Ap = Ac[pi][:, pi].tocsr()
    nt = nc // 16
    t = np.arange(nt)
    even = (((t // 4) + (t % 4)) & 1) == 0
    # the packed operators as sparse matrices again
    def unpack(tm, kc, vals):
        RT, KS = kc.shape
        rows = (tm[:, None, None, None] * 16 + np.arange(16)[None, None, None, :]) + np.zeros((1, KS, 4, 1), dtype=np.int64)
        cols = kc[:, :, None, None] + np.arange(4)[None, None, :, None] + np.zeros((1, 1, 1, 16), dtype=np.int64)
        v = vals.reshape(RT, KS, 4, 16)                      # [rt, ks, column in group, row in tile]
        return sp.csr_matrix((v.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(nc, nc))
    S, F, G, Hb = (unpack(*pk) for pk in packed)
    assert np.array_equal(np.sort(packed[0][0]), np.nonzero(even)[0])
    E = np.nonzero(np.repeat(even, 16))[0]
    O = np.nonzero(~np.repeat(even, 16))[0]
    b = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
    x = spla.spsolve(Ap.tocsc(), b)
    bp = b - F @ b
    xe = np.linalg.solve(S[E][:, E].toarray(), bp[E])
    xx = np.zeros_like(b)
    xx[E] = xe
    xx[O] = (G @ b - Hb @ xx)[O]
    assert np.linalg.norm(xx - x) / np.linalg.norm(x) < 1e-12
    assert hierarchy.reference_coarse_eo(P, Ac, 128, 64) is None          # other aggregate size
    Pbad = sp.csr_matrix(P)[:, : nc - 8]
    assert hierarchy.reference_coarse_eo(Pbad, Ac[: nc - 8][:, : nc - 8], 128, 32) is None

