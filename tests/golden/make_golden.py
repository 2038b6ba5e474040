#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only; needs
/root/reference, which never travels to the GPU box).

What can be taken from the reference itself
-------------------------------------------
``multigrid.py`` (and everything that imports it) cannot be imported here: ``pyamg`` is not
installed (ordinary ModuleNotFoundError).  ``utils.py`` and ``matrix.py`` import fine, so the
fixtures below are produced by the REFERENCE's own

* ``utils.one_defl_Hutch_step``          (probe draw, deflation projection, Pperm^T, vdot,
                                          MLMC difference estimate),
* ``utils.deflation_pre_computations``   (U = Pperm g3 V sgn, tr1),
* ``utils.flopsV_manual``, ``utils.trace_params_from_params``,
* ``matrix.loadMatrix``                  (128^2 only: the 16^2 branch raises on SciPy>=1.14),

called with a duck-typed ``mg_solver`` whose ``solve`` is an exact sparse-LU solve (the
converged MG-FGMRES solve equals it to ~1e-12).  The hierarchy objects handed to
those functions come from oracle/ref_path.py (restatement of multigrid.py:100-345), with the
16^2 test vectors stored in the fixture so the MLMC values are reproducible.

Known-answer values copied from the reference's text: the "exact trace" comment,
gateway.py:100-104.
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

os.environ["OMP_NUM_THREADS"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import utils as ref_utils          # the reference's utils.py   # noqa: E402
import matrix as ref_matrix        # the reference's matrix.py  # noqa: E402
from oracle import ref_path as rp  # noqa: E402


class DuckSolver:
    """What utils.one_defl_Hutch_step / deflation_pre_computations touch on mg_solver."""

    class _ML:
        pass

    def __init__(self, levels, coarsest_inv, skip_level):
        self.ml = DuckSolver._ML()
        self.ml.levels = levels
        self.coarsest_inv = np.matrix(coarsest_inv)
        self.skip_level = skip_level
        self.timer = ref_utils.CustomTimer()
        self.level_nr = 0
        self.num_iters = 0
        self.solve_tol = 0.1
        self.x = None
        self._lu = {}

    def solve(self, A, b, tol):
        key = A.shape[0]
        if key not in self._lu:
            self._lu[key] = spla.splu(sp.csc_matrix(self.ml.levels[self.level_nr].A))
        self.x = self._lu[key].solve(np.asarray(b, dtype=np.complex128))
        self.num_iters = 1


def cplx_list(a):
    return [[float(np.real(v)), float(np.imag(v))] for v in np.atleast_1d(a)]


def silence():
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO())


def main():
    out = {}
    # ---- RNG known answers -------------------------------------------------
    np.random.seed(123456)
    st = np.random.get_state()
    rs = np.random.RandomState()
    rs.set_state(st)
    out["mt19937_seed123456_first_words"] = [int(v) for v in
                                             rs.randint(0, 2 ** 32, size=8, dtype=np.uint64)]
    np.random.seed(123456)
    x = np.random.randint(2, size=32768)
    out["probe0_n32768_sum"] = int((2 * x - 1).sum())
    out["probe0_n32768_first16"] = [int(v) for v in (2 * x - 1)[:16]]

    # ---- reference text KAT -------------------------------------------------------------
    out["gateway_exact_trace_128"] = [-8.748242701374695, 50.215154098005584]   # gateway.py:104

    # ---- 128^2 ------------------------------------------------------------------------
    mass128 = -0.1320
    cwd = os.getcwd()
    os.chdir(REF)
    with silence():
        A128 = sp.csr_matrix(ref_matrix.loadMatrix('schwinger128.mat',
                                                   {'mass': mass128, 'problem_name': 'schwinger'}))
    os.chdir(cwd)
    A128o = rp.load_matrix(os.path.join(REF, 'schwinger128.mat'), mass128)
    d = abs(A128 - A128o)
    out["loadMatrix128_matches_restatement_maxdiff"] = float(d.max()) if d.nnz else 0.0
    n = A128.shape[0]
    params128 = {'use_permuted': True, 'latt_dims': [128, 128], 'x_displacement': 2,
                 'function_params': {'tol': 1e-12}, 'defl_type': 'exact'}
    lev0 = rp.OLevel()
    lev0.A = A128
    sign = np.ones(n)
    sign[n // 2:] = -1
    lev0.g3 = sp.diags([sign], [0])
    lev0.perm_shift = 512
    lev0.Pperm = rp.pperm_matrix(n, 512)
    duck = DuckSolver([lev0], np.eye(2), False)
    # plain probes (k = 0): hierarchy- and eigenvector-independent
    np.random.seed(123456)
    plain = []
    with silence():
        for _ in range(6):
            e, _ = ref_utils.one_defl_Hutch_step(A128, None, duck, params128, "hutchinson", 0,
                                                 None, None)
            plain.append(e)
    out["hutch128_plain_seed123456"] = cplx_list(plain)
    # deflated probes (k = 8, tol 1e-9) through the reference's own deflation code
    with silence():
        Ux, tr1 = ref_utils.deflation_pre_computations(A128, 8, 1e-9, "hutchinson", duck.timer,
                                                       params128, duck)
    out["defl128_tr1"] = cplx_list(tr1)[0]
    np.random.seed(123456)
    defl = []
    with silence():
        for _ in range(16):
            e, _ = ref_utils.one_defl_Hutch_step(A128, None, duck, params128, "hutchinson", 8,
                                                 Ux, None)
            defl.append(e)
    out["hutch128_deflated_k8_seed123456"] = cplx_list(defl)
    # exact traces by LU
    out["exact_trace_128_perm"] = cplx_list(rp.exact_trace_inverse(A128, lev0.Pperm.transpose()))[0]
    out["exact_trace_128_plain"] = cplx_list(rp.exact_trace_inverse(A128))[0]

    # ---- 16^2 ---------------------------------------------------------------------------
    mass16 = -1.00690114 * 0.99
    A16 = rp.load_matrix(os.path.join(REF, 'schwinger16.mat'), mass16)
    n16 = A16.shape[0]
    out["exact_trace_16_plain"] = cplx_list(rp.exact_trace_inverse(A16))[0]
    params16 = {'use_permuted': False, 'latt_dims': [16, 16], 'x_displacement': 0,
                'function_params': {'tol': 1e-12}, 'defl_type': 'exact',
                'test_vectors_type': 'EVs'}
    dof, aggrs = [2, 4, 4], [4, 4, 4]
    levels16, cinv16, tv16 = rp.mg_setup(A16, dof, aggrs, 3, 'high', params16)
    np.savez_compressed(os.path.join(HERE, "schwinger16_testvectors.npz"),
                        tv0=tv16[0], tv1=tv16[1])
    duck16 = DuckSolver(levels16, cinv16, False)
    np.random.seed(123456)
    plain16 = []
    with silence():
        for _ in range(32):
            e, _ = ref_utils.one_defl_Hutch_step(A16, None, duck16, params16, "hutchinson", 0,
                                                 None, None)
            plain16.append(e)
    out["hutch16_plain_seed123456"] = cplx_list(plain16)
    with silence():
        Ux16, tr1_16 = ref_utils.deflation_pre_computations(A16, 8, 1e-9, "hutchinson",
                                                            duck16.timer, params16, duck16)
    out["defl16_tr1"] = cplx_list(tr1_16)[0]
    np.random.seed(123456)
    d16 = []
    with silence():
        for _ in range(16):
            e, _ = ref_utils.one_defl_Hutch_step(A16, None, duck16, params16, "hutchinson", 8,
                                                 Ux16, None)
            d16.append(e)
    out["hutch16_deflated_k8_seed123456"] = cplx_list(d16)
    # MLMC difference estimates through the reference's own utils (levels 0 and 1, no skip)
    for lvl in (0, 1):
        np.random.seed(4242 + lvl)
        outp = {'results': [{'function_iters': 0} for _ in range(3)]}
        vals = []
        with silence():
            for _ in range(8):
                e, _ = ref_utils.one_defl_Hutch_step(levels16[lvl].A, levels16[lvl + 1].A, duck16,
                                                     params16, "mlmc", 0, None, None, lvl, outp,
                                                     levels16[lvl].P, levels16[lvl].R)
                vals.append(e)
        out["mlmc16_level%d_seed%d" % (lvl, 4242 + lvl)] = cplx_list(vals)
    # permuted 16^2 variant (shift = 16*2*1) with level skipping, exercises Bblock_perm
    params16p = dict(params16, use_permuted=True, x_displacement=1)
    levels16p, cinv16p, _ = rp.mg_setup(A16, dof, aggrs, 3, 'high', params16p, testvectors=tv16)
    duck16p = DuckSolver(levels16p, cinv16p, True)
    np.random.seed(777)
    outp = {'results': [{'function_iters': 0} for _ in range(3)]}
    vals = []
    with silence():
        for _ in range(8):
            e, _ = ref_utils.one_defl_Hutch_step(levels16p[0].A, levels16p[2].A, duck16p, params16p,
                                                 "mlmc", 0, None, None, 0, outp, levels16p[0].P,
                                                 levels16p[0].R, levels16p[1].P, levels16p[1].R)
            vals.append(e)
    out["mlmc16_perm_skip_level0_seed777"] = cplx_list(vals)

    # ---- plumbing goldens from the reference's utils --------------------------------------
    class _L:
        def __init__(self, nnz):
            self.A = type("M", (), {"nnz": nnz})()

    class _S:
        smooth_iters = 2
    lv = [_L(294912), _L(294904), _L(98304), _L(24576)]
    out["flopsV_manual"] = {"(4,0)": ref_utils.flopsV_manual(4, lv, 0, _S()),
                            "(0,0)": ref_utils.flopsV_manual(0, lv, 0, _S()),
                            "(2,2)": ref_utils.flopsV_manual(2, lv, 2, _S()),
                            "(1,1)": ref_utils.flopsV_manual(1, lv, 1, _S())}
    full = {'function_tol': 1e-12, 'trace_tol': 1e-2, 'max_nr_levels': 4,
            'matrix_params': {'problem_name': 'schwinger', 'mass': -0.132},
            'nr_deflat_vctrs': 8, 'mlmc_deflat_vctrs': [0, 0, 0], 'defl_eigvs_tol_Hutch': 1e-9,
            'defl_eigvs_tol_MLMC': 0.1, 'diff_lev_op_tol': 1e-3, 'defl_type': 'exact',
            'coarsest_level_directly': True, 'accuracy_mg_eigvs': 'high', 'aggrs': [16, 4, 4],
            'dof': [2, 8, 8, 8], 'mlmc_levels_to_skip': [1], 'use_permuted': True,
            'latt_dims': [128, 128], 'x_displacement': 2, 'check_quality_MG': False,
            'test_vectors_type': 'EVs'}
    out["trace_params_mlmc"] = ref_utils.trace_params_from_params(full, "mlmc")
    out["trace_params_hutchinson"] = ref_utils.trace_params_from_params(full, "hutchinson")

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
