"""Helpers of the reference's ``utils.py`` (same names and call signatures), with the probe
body running on the GPU engine.  Reference lines are cited per function."""
import os
import time

import numpy as np
from scipy.sparse.linalg import eigsh

from . import dist as _dist
from .engine import MODE_HUTCHINSON, MODE_LEVEL, MODE_MLMC, MODE_MLMC_SKIP, EngineError


# ----------------------------------------------------------------------------------------
# reporting / plumbing
# ----------------------------------------------------------------------------------------
def flopsV_manual(bare_level, levels_info, level_id, mg_solver):
    """utils.py:19-31: nnz-weighted work model of one cycle (kept as is, not "fixed")."""
    total = 0
    last = len(levels_info) - 2
    lvl = level_id
    while True:
        weight = 2 * mg_solver.smooth_iters + (2 if lvl == bare_level else 1)
        total += weight * levels_info[lvl].A.nnz
        if lvl == last:
            return total
        lvl += 1


def print_post_results(A, params, result, example):
    """utils.py:36-69."""
    if example not in ("mlmc", "hutchinson"):
        raise Exception("Value for parameter <example> not available.")
    print(" -- matrix : " + params['matrix'])
    print(" -- matrix size : " + str(A.shape[0]) + "x" + str(A.shape[1]))
    print(" -- tr(A^{-1}) = " + str(result['trace']))
    print(" -- total MG complexity = " + str(result['total_complexity'] / (1.0e+6)) + " MFLOPS")
    if example == "mlmc":
        print(" -- std dev = ---")
        for i in range(result['nr_levels']):
            lev = result['results'][i]
            print(" -- level : " + str(i))
            print(" \t-- number of estimates = " + str(lev['nr_ests']))
            print(" \t-- function iters = " + str(lev['function_iters']))
            print(" \t-- trace = " + str(lev['ests_avg']))
            print(" \t-- std dev = " + str(lev['ests_dev']))
            print(" \t-- var = " + str(lev['ests_dev'] * lev['ests_dev']))
            print("\t-- level MG complexity = " + str(lev['level_complexity'] / (1.0e+6)) + " MFLOPS")
    else:
        print(" -- std dev = " + str(result['std_dev']))
        print(" -- var = " + str(result['std_dev'] * result['std_dev']))
        print(" -- number of estimates = " + str(result['nr_ests']))
        print(" -- function iters = " + str(result['function_iters']))


_COMMON_KEYS = ('max_nr_levels', 'nr_deflat_vctrs', 'defl_eigvs_tol_Hutch', 'accuracy_mg_eigvs',
                'aggrs', 'dof', 'use_permuted', 'latt_dims', 'x_displacement', 'check_quality_MG',
                'test_vectors_type')
_MLMC_KEYS = ('mlmc_deflat_vctrs', 'defl_eigvs_tol_MLMC', 'diff_lev_op_tol', 'defl_type',
              'coarsest_level_directly', 'mlmc_levels_to_skip')
# build-only options (all optional; reference presets do not carry them)
_BUILD_KEYS = ('batch', 'device', 'engines', 'cache_dir', 'report_path', 'probe_type', 'solver_cfg', 'use_solver_hierarchy', 'mg_testvectors',
               'solver_testvectors', 'deflation_eigenpairs', 'ref_cycle_post', 'ref_cycle_k', 'ref_smoother',
               'solver_restart', 'stochastic_coarsest', 'stop_factor', 'ref_direct_max_n', 'ref_coarsest',
               'ref_coarse_dofs', 'setup_eigs', 'defer_coarse_levels',
               'verbose', 'probe_rounds_max')


def trace_params_from_params(params, example):
    """utils.py:73-125: whitelist copy into the dictionary the estimators read."""
    if example not in ("mlmc", "hutchinson"):
        raise Exception("Value for parameter <example> not available.")
    tp = {'function_params': {'tol': params['function_tol']},
          'tol': params['trace_tol'],
          'max_nr_ests': 100000,
          'problem_name': params['matrix_params']['problem_name']}
    for key in _COMMON_KEYS:
        tp[key] = params[key]
    if example == "mlmc":
        for key in _MLMC_KEYS:
            tp[key] = params[key]
    else:
        tp['defl-type'] = params['defl_type']      # key spelled as in utils.py:113
    for key in _BUILD_KEYS:
        if key in params:
            tp[key] = params[key]
    return tp


class CustomTimer:
    """utils.py:366-445: non-reentrant wall-clock buckets.  On this build the buckets are
    additionally fed from HIP-event timings of the engine (MG.sync_timer)."""
    _PARTS = ("mvm", "defl", "P", "R", "mg_setup", "defl_setup", "axpy")

    def __init__(self):
        self.on = 0
        self.reset()

    def reset(self):
        for part in self._PARTS:
            setattr(self, part, 0.0)
        self.dots = 0.0      # Krylov inner products: untimed in the reference
        self.tbuff = 0.0

    def start(self, part):
        if self.on == 1:
            raise Exception("Can't turn timer on, it's already timing")
        self.on = 1
        self.tbuff = time.time()

    def end(self, part):
        if self.on == 0:
            raise Exception("Can't turn timer off, it's already down")
        self.on = 0
        elapsed = time.time() - self.tbuff
        if part not in self._PARTS:
            raise Exception("Uknown part to time")
        setattr(self, part, getattr(self, part) + elapsed)

    def __str__(self):
        acc = self.mvm + self.defl + self.P + self.R + self.mg_setup + self.defl_setup
        lines = ["", "Timings specific to computations:",
                 " -- matrix-vector multiplications : " + str(self.mvm),
                 " -- deflations : " + str(self.defl),
                 " -- applications of P : " + str(self.P),
                 " -- applications of R : " + str(self.R),
                 " -- applications of axpy : " + str(self.axpy),
                 " -- accumulated time : " + str(acc), ""]
        return "\n".join(lines)


def _engines(mg_solver):
    engs = getattr(mg_solver, "engines", None)
    if engs:
        return engs
    return [mg_solver.engine] if getattr(mg_solver, "engine", None) is not None else []


# ----------------------------------------------------------------------------------------
# deflation (setup-time, host)                                          utils.py:130-201
# ----------------------------------------------------------------------------------------
def deflation_pre_computations(A, nr_deflat_vctrs, tolx, method, timer, params, mg_solver,
                               lop=None, level_nr=0):
    if method not in ("hutchinson", "mlmc"):
        raise Exception("unknown deflation method")
    if nr_deflat_vctrs <= 0:
        if method == "hutchinson":
            for eng in _engines(mg_solver):
                eng.set_deflation(None)
            return (None, 0.0)
        for eng in _engines(mg_solver):
            eng.set_level_deflation(level_nr, None)
        return (None, None, 0.0)

    lev0 = mg_solver.ml.levels[0]
    if method == "hutchinson":
        pre = params.get("deflation_eigenpairs") if hasattr(params, "get") else None
        if pre is not None:
            Sy, Vx = np.array(pre[0], dtype=float), np.array(pre[1], dtype=np.complex128)
        else:
            from . import cache as _cache
            cdir = _cache.cache_dir(params)
            ckey = _cache.matrix_key(A, {"k": nr_deflat_vctrs, "tol": tolx}) if cdir else None
            hit = _cache.load(cdir, "defl", ckey) if cdir else None
            found = getattr(mg_solver, "_device_defl", {}).get((int(nr_deflat_vctrs), float(tolx)))
            if hit is not None:
                Sy, Vx = hit["S"], hit["V"]
            elif found is not None:
                # computed on the GPU during MG.setup, while the host built the coarse levels
                Sy, Vx = found
            else:
                device = (getattr(mg_solver, "_have_solver_hier", False) and nr_deflat_vctrs <= 32
                          and getattr(mg_solver, "_solver_cfg_built", None) is not None
                          and mg_solver._solver_cfg_built.get("setup") == "device"
                          and params.get("setup_eigs", os.environ.get("SW_SETUP_EIGS", "device")) == "device")
                if device:
                    # eigsh(gamma_3 A, k, sigma=0) by block subspace iteration on the GPU (utils.py:137-140)
                    Sy, Vx = _dist.default_comm().compute_on_root(
                        lambda: mg_solver.device_eigenpairs(nr_deflat_vctrs, tolx, hermitian=True))
                else:
                    Q = (lev0.g3 * A).tocsc()                               # utils.py:137-140
                    # (with several ranks: rank 0's eigenpairs everywhere, see dist.compute_on_root)
                    Sy, Vx = _dist.default_comm().compute_on_root(
                        lambda: eigsh(Q, k=nr_deflat_vctrs, which='LM', tol=tolx, sigma=0.0))
            if hit is None and cdir:
                _cache.save(cdir, "defl", ckey, {"S": Sy, "V": Vx})
    else:
        mg_solver.solve_tol = params['diff_lev_op_tol']                 # utils.py:142-143
        Sy, Vx = _dist.default_comm().compute_on_root(
            lambda: eigsh(lop, k=nr_deflat_vctrs, which='LM', tol=tolx))
    sgn = np.where(Sy > 0, 1.0, -1.0)
    Sabs = Sy * sgn
    Ux = Vx * sgn[None, :]
    if method == "hutchinson":
        Ux = lev0.g3 * Ux
        if params['use_permuted']:
            Ux = lev0.Pperm * Ux
    else:
        Vx = mg_solver.ml.levels[level_nr].g3 * Vx
        if getattr(mg_solver, "coarse_eo", None) is not None and level_nr >= 1:
            raise Exception("ref_coarsest = 'eo' keeps the coarse level in tile order on the GPU: MLMC-level "
                            "deflation vectors at level %d are not supported in that mode" % level_nr)
        for eng in _engines(mg_solver):
            # the GPU probe body projects with these vectors (utils.py:260-266)
            eng.set_level_deflation(level_nr, np.asarray(Vx))

    if os.getenv('OMP_NUM_THREADS') is None:                            # utils.py:161-164
        raise Exception("Run : << export OMP_NUM_THREADS=N >>")
    mg_solver.solve_tol = params['function_params']['tol']

    overlap = np.dot(Ux.transpose().conjugate(), Vx)
    if method == "hutchinson":
        tr1 = np.sum(np.diag(overlap) / Sabs)                           # utils.py:173,191
        for eng in _engines(mg_solver):
            eng.set_deflation(np.asarray(Ux))
        return (Ux, tr1)
    defl_type = params['defl_type']
    if defl_type == "exact":
        tr1 = np.sum(np.diag(overlap) * Sabs)                           # utils.py:176
    elif defl_type == "inexact_01":
        Vbuff = np.zeros_like(Vx)
        for i in range(nr_deflat_vctrs):
            Vbuff[:, i] = mg_solver.diff_op(Vx[:, i].copy())
            print('.', end='', flush=True)
        tr1 = np.trace(np.dot(Vx.transpose().conjugate(), Vbuff))
    elif defl_type == "inexact_02":
        raise Exception("deflation type inexact_02 under construction")
    elif defl_type == "inexact_03":
        tr1 = 0.0
    else:
        raise Exception("unknown deflation type")
    return (Vx, Ux, tr1)


# ----------------------------------------------------------------------------------------
# probes
# ----------------------------------------------------------------------------------------
def draw_probes(count, n, kind="z2"):
    """`count` probes from the GLOBAL NumPy stream as int8 codes, shape (count, n).

    kind "z2" (the reference): identical to `count` calls of np.random.randint(2, size=n)
    (utils.py:213-215), entries +-1.  kind "z4" (build-only option, BASELINE config 1; NOT in the
    reference): one draw of np.random.randint(4) per entry, 0,1,2,3 -> 1, i, -1, -i, encoded as
    +1, +2, -1, -2."""
    if kind == "z2":
        bits = np.random.randint(2, size=(count, n))
        return (2 * bits - 1).astype(np.int8)
    if kind == "z4":
        q = np.random.randint(4, size=(count, n))
        return np.array([1, 2, -1, -2], dtype=np.int8)[q]
    raise Exception("unknown probe type")


def probes_as_complex(probes):
    """int8 probe codes -> the complex vectors they stand for."""
    p = np.asarray(probes)
    return np.where(np.abs(p) == 2, 1j * (p // 2), p).astype(np.complex128)


def probe_batch(mg_solver, params, method, probes, level=0):
    """Evaluate one batch of probes on the GPU: returns (ests, iters_fine, iters_coarse).

    hutchinson: e = x^H A^-1 Pperm^T (x - U U^H x)            utils.py:210-250
    mlmc      : e = x^H A_f^-1 C x - x^H P A_c^-1 R C x        utils.py:252-361
    (deflation vectors and permutation were registered with the engine at setup)."""
    engs = _engines(mg_solver)
    if not engs:
        raise EngineError("no GPU engine attached (run MG.setup first)")
    tol = params['function_params']['tol']
    n = mg_solver.ml.levels[level].A.shape[0]
    maxiter = n if n < 1000 else 1000
    mode = _probe_mode(mg_solver, method, level)
    probes = np.asarray(probes)
    nb = probes.shape[0]
    if len(engs) == 1 or nb < 2 * 64:
        return engs[0].hutch_batch(mode, level, probes, tol, maxiter)
    # several engine handles = several HIP streams on the same GPU: the sub-batches overlap
    # (MFMA-bound coarse kernels of one with HBM-bound fine kernels of the other)
    from concurrent.futures import ThreadPoolExecutor
    parts = np.array_split(np.arange(nb), len(engs))
    with ThreadPoolExecutor(max_workers=len(engs)) as pool:
        futs = [pool.submit(eng.hutch_batch, mode, level, probes[idx], tol, maxiter)
                for eng, idx in zip(engs, parts) if len(idx)]
        res = [f.result() for f in futs]
    return tuple(np.concatenate([r[k] for r in res]) for k in range(3))


def _probe_mode(mg_solver, method, level):
    if method == "hutchinson":
        return MODE_HUTCHINSON
    if method == "mlmc":
        return MODE_MLMC_SKIP if (mg_solver.skip_level and level == 0) else MODE_MLMC
    if method == "level":
        return MODE_LEVEL
    raise Exception("unknown method")


def probe_batch_generated(mg_solver, params, method, level, first_probe, count, kind="z2",
                          prefetch=None, ready=None):
    """Like probe_batch, for the probes [first_probe, first_probe + count) of the stream the
    engines were handed with Engine.stream_set: each engine GENERATES its contiguous share in
    HBM (k_mt_generate, bit-exact with np.random.randint, utils.py:213-216,255-258) and
    evaluates it; only the 16-byte estimates come back.

    Generation is asynchronous on the engines' generation streams.  `prefetch` = (first_probe, count)
    of the batch expected NEXT: its probes are queued into the engines' other slot before this batch is
    solved, so they are drawn while the solve runs; `ready` = the (first_probe, count, slot) a previous
    call prefetched (skips this batch's own generation when it matches).  Returns
    (ests, iters_fine, iters_coarse, prefetched) with prefetched = (first, count, slot) or None."""
    engs = _engines(mg_solver)
    if not engs:
        raise EngineError("no GPU engine attached (run MG.setup first)")
    tol = params['function_params']['tol']
    n = mg_solver.ml.levels[level].A.shape[0]
    maxiter = n if n < 1000 else 1000
    mode = _probe_mode(mg_solver, method, level)
    ne = len(engs) if count >= 2 * 64 else 1

    def shares(cnt):
        return [(k * cnt) // ne for k in range(ne + 1)]

    slot = 0
    have = ready is not None and ready[0] == first_probe and ready[1] == count
    if have:
        slot = ready[2]
    bounds = shares(count)
    if not have:
        for k in range(ne):
            if bounds[k + 1] > bounds[k]:
                engs[k].probes_generate(slot, level, bounds[k + 1] - bounds[k],
                                        (first_probe + bounds[k]) * n, kind)
    prefetched = None
    if prefetch is not None and prefetch[1] > 0 and (ne == (len(engs) if prefetch[1] >= 2 * 64 else 1)):
        nslot = 1 - slot
        nb2 = shares(prefetch[1])
        for k in range(ne):
            if nb2[k + 1] > nb2[k]:
                engs[k].probes_generate(nslot, level, nb2[k + 1] - nb2[k], (prefetch[0] + nb2[k]) * n, kind)
        prefetched = (prefetch[0], prefetch[1], nslot)

    def run(eng):
        eng.probes_select(slot)
        eng.hutch_run(mode, level, tol, maxiter)
        return eng.hutch_fetch()

    active = [engs[k] for k in range(ne) if bounds[k + 1] > bounds[k]]
    if len(active) == 1:
        res = [run(active[0])]
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(active)) as pool:
            res = list(pool.map(run, active))
    return tuple(np.concatenate([r[k] for r in res]) for k in range(3)) + (prefetched,)


def one_defl_Hutch_step(Af, Ac, mg_solver, params, method, nr_deflat_vctrs, Vx, Ux, i=0,
                        output_params=None, P=None, R=None, Pn=None, Rn=None):
    """utils.py:207-361, one probe.  The probe comes from the global NumPy stream exactly as in
    the reference; the arithmetic runs on the GPU.  MLMC-level deflation vectors (Vx with
    method == "mlmc") are applied on the host before the batch call."""
    n = Af.shape[0]
    if method == "hutchinson":
        probes = draw_probes(1, n)
        mg_solver.level_nr = 0
        e, itf, _ = probe_batch(mg_solver, params, "hutchinson", probes, 0)
        mg_solver.num_iters = int(itf[0])
        itrs = int(itf[0])
        est = e[0]
    elif method == "mlmc":
        if nr_deflat_vctrs > 0 and params['defl_type'] not in ("exact", "inexact_01"):
            if params['defl_type'] == "inexact_02":
                raise Exception("deflation type inexact_02 under construction")
            if params['defl_type'] == "inexact_03":
                raise Exception("deflation type inexact_03 is not available on the GPU probe path")
            raise Exception("unknown deflation type")
        # with nr_deflat_vctrs > 0 the projection x0 - V V^H x0 (utils.py:266) uses the vectors that
        # deflation_pre_computations registered with the engine for level i
        probes = draw_probes(1, n)
        mg_solver.level_nr = i
        e, itf, itc = probe_batch(mg_solver, params, "mlmc", probes, i)
        lc = i + 2 if (mg_solver.skip_level and i == 0) else i + 1
        output_params['results'][i]['function_iters'] += int(itf[0])
        output_params['results'][lc]['function_iters'] += int(itc[0])
        itrs = 0
        est = e[0]
    else:
        raise Exception("unknown method")
    print('.', end='', flush=True)
    return (est, itrs)
