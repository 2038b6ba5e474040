#!/usr/bin/env python3
"""GPU tuning aid: what the even-odd smoother polynomial of the lattice level is fitted ON.

The shipped fit is the GMRES(nu) residual polynomial of the Schur complement S on ONE random start
vector.  Inside the cycle the smoother never sees such a vector: it sees what the coarse correction
leaves, r = b - S (P A_c^-1 R b)_even.  This script fits the polynomial on that class of vectors
(one vector by Arnoldi, several by least squares) and on plain random vectors (control), installs each
weight set with sw_set_eo_smoother and reports the residual after 8 outer iterations, the strict-mode
iteration count and the batch rate.  Results: profiles/r04_ab_sessions.txt."""
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np  # noqa: E402

from deflatedmlmc_schwinger_amd import gateway, matrix, utils, setup_gpu  # noqa: E402
from deflatedmlmc_schwinger_amd import hierarchy as H  # noqa: E402
from deflatedmlmc_schwinger_amd.engine import MODE_HUTCHINSON, ProbeStream  # noqa: E402
from deflatedmlmc_schwinger_amd.multigrid import MG, SOLVER_HID  # noqa: E402


def leja_weights(theta):
    theta = list(theta)
    ordered = [max(theta, key=abs)]
    theta.remove(ordered[0])
    while theta:
        nxt = max(theta, key=lambda t: np.prod([abs(t - o) for o in ordered]))
        ordered.append(nxt)
        theta.remove(nxt)
    return 1.0 / np.array(ordered, dtype=np.complex128)


def ls_weights(S, D, starts, degree):
    """min over p (p(0) = 1, degree `degree`) of sum_i |p(S) r_i|^2: monomials of S / D."""
    cols, rhs = [], []
    for r in starts:
        K = []
        v = r
        for _ in range(degree):
            v = (S @ v) / D
            K.append(v)
        cols.append(np.stack(K, axis=1))
        rhs.append(r)
    K = np.concatenate(cols, axis=0)
    b = np.concatenate(rhs)
    c = np.linalg.lstsq(K, b, rcond=None)[0]          # p(z) = 1 - sum_k c_k z^k, z = S / D
    poly = np.concatenate([-c[::-1], [1.0]])          # highest power first
    roots = np.roots(poly) * D                        # roots in terms of S
    return leja_weights(roots), float(np.linalg.norm(b - K @ c) / np.linalg.norm(b))


def main():
    nb = int(os.environ.get("SW_NB", "256"))
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    params['solver_cfg'] = dict(H.TUNED_SOLVER_CFG_128)
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "hutchinson")
    mg = MG(A)
    with contextlib.redirect_stdout(io.StringIO()):
        mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
                 acc_eigvs=tp['accuracy_mg_eigvs'], sys_type=tp['problem_name'], params=tp)
        utils.deflation_pre_computations(A, 8, 1e-9, "hutchinson", mg.timer, tp, mg)
    eng = mg.engine
    hid = SOLVER_HID
    L, mass = mg.lattice[0], mg.lattice[1]
    n = 2 * L * L
    nu = int(params['solver_cfg']["cycle"][0][1])
    S = setup_gpu._EngineSchur(eng, hid, L, mass)
    E = S.E
    D = S.D
    probes = ProbeStream(123456).rademacher(nb, n)
    eng.probes_upload(0, probes)
    rng = np.random.default_rng(7)
    Btest = rng.standard_normal((64, n)) + 1j * rng.standard_normal((64, n))

    def corrected(b):
        u = np.zeros(n, dtype=np.complex128)
        u[E] = b
        ec = eng.solve(hid, 1, eng.restrict(hid, 0, u), 1e-12)[0]
        return b - S @ eng.prolong(hid, 0, ec)[E]

    def rand_even(seed):
        r = np.random.default_rng(seed)
        return r.standard_normal(E.size) + 1j * r.standard_normal(E.size)

    def measure(label, w):
        w = np.asarray(w, dtype=np.complex128)
        eng.set_eo_smoother(hid, 0, w)
        out = {"label": label, "degree": int(w.size)}
        for its in (4, 8):
            eng.set_option("stop_factor", 1e-30)
            try:
                _, _, rel = eng.solve(hid, 0, Btest, 1e-12, its)
                out["log10_relres_after_%d" % its] = round(float(np.mean(np.log10(rel))), 3)
                out["worst_after_%d" % its] = float(rel.max())
            except Exception as exc:                       # iteration cap reported as an error: read no number
                out["solve_error"] = str(exc)[:100]
        for sf, key in ((0.1, "strict"), (1.0, "refstop")):
            eng.set_option("stop_factor", sf)
            eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000)
            eng.sync()
            dt = (time.perf_counter() - t0) / 5
            _, itf, _ = eng.hutch_fetch()
            out[key] = {"iters": int(itf.max()), "probes_per_s": round(nb / dt)}
        print(json.dumps(out), flush=True)

    base = H.weights_from_hessenberg(eng.setup_arnoldi(hid, 0, 1, nu))
    measure("device Arnoldi, random start (shipped)", base)
    measure("host Arnoldi, random start", H.smoother_weights(S, nu))
    for seed in (2024, 5):
        measure("host Arnoldi, coarse-corrected start, seed %d" % seed,
                H.smoother_weights(S, nu, seed=seed, project=corrected))
    cs = [corrected(rand_even(100 + i)) for i in range(8)]
    for k in (1, 4, 8):
        w, fit = ls_weights(S, D, cs[:k], nu)
        measure("least squares on %d coarse-corrected vectors (fit residual %.3e)" % (k, fit), w)
    w, fit = ls_weights(S, D, [rand_even(200 + i) for i in range(8)], nu)
    measure("least squares on 8 random vectors (fit residual %.3e)" % fit, w)
    for deg in (nu - 1, nu - 2):
        w, fit = ls_weights(S, D, cs, deg)
        measure("degree %d, least squares on 8 coarse-corrected vectors (fit residual %.3e)" % (deg, fit), w)
        measure("degree %d, Arnoldi random start" % deg, H.smoother_weights(S, deg))


if __name__ == "__main__":
    main()
