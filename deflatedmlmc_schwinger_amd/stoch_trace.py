"""``hutchinson`` and ``mlmc`` of the reference's stoch_trace.py on the MI355X engine.

Same call surface and result dictionaries.  The reference evaluates one probe at
a time; here the probes of a round are one multi-RHS batch on the GPU (and, with several
ranks, a contiguous slice of the round per GPU).  The probes come from the same seeded
global NumPy stream in the same order, and the sequential stopping rule is replayed over the
gathered per-probe values, so ``trace``, ``std_dev`` and ``nr_ests`` are what the one-by-one
loop would produce with the same per-probe values.
"""
import time
from math import sqrt

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.linalg import LinearOperator

from . import dist as _dist
from .engine import ProbeStream
from .multigrid import MG
from .utils import (_engines, deflation_pre_computations, draw_probes, flopsV_manual, probe_batch,
                    probe_batch_generated)

DEFAULT_BATCH = 256
NR_ROUGH_PROBES = 5


def _stats(values):
    """Mean and population deviation exactly as stoch_trace.py:143-145 writes them."""
    count = len(values)
    avg = np.sum(values) / count
    dev = sqrt(np.sum(np.square(np.abs(values - avg))) / count)
    return avg, dev


def first_stop_index(ests, first_new, level_tol, min_index=5):
    """The first index i >= first_new at which the reference's loop would break,
    `i >= min_index and dev_i / sqrt(i + 1) < level_tol` with (avg_i, dev_i) = _stats(ests[:i + 1])
    (stoch_trace.py:143-154 / 394-406), or None -- without the reference's O(N^2) re-evaluation of the
    statistics per probe: a running-sum screen (one pass, on estimates shifted by a pivot so the
    variance does not cancel) marks the indices whose error estimate is below level_tol or within a
    relative 1e-9 of it -- many orders above the screen's rounding error --, and only those are
    re-evaluated with the reference's own two-pass formula, in order.  Returns (index, avg, dev) of the
    break, or None when the loop runs on."""
    ests = np.asarray(ests, dtype=np.complex128)
    n = ests.size
    lo = max(int(first_new), int(min_index))
    if n == 0 or lo >= n:
        return None
    pivot = np.mean(ests[:min(n, 64)])
    d = ests - pivot
    cnt = np.arange(1, n + 1, dtype=np.float64)
    m1 = np.cumsum(d) / cnt
    m2 = np.cumsum(d.real * d.real + d.imag * d.imag) / cnt
    var = np.maximum(m2 - (m1.real * m1.real + m1.imag * m1.imag), 0.0)
    err = np.sqrt(var / cnt)
    cand = np.flatnonzero(err[lo:] < level_tol * (1.0 + 1e-9) + 1e-300) + lo
    for i in cand:
        avg, dev = _stats(ests[:i + 1])
        if dev / sqrt(i + 1) < level_tol:
            return int(i), avg, dev
    return None


class HostProbes:
    """Probe source for evaluators that take the probes themselves (int8, (k, n)): this rank's
    slice of a round is produced on the host by the C MT19937 stream at its stream positions
    (jump polynomial, no walk over the other ranks' draws)."""

    def __init__(self, evaluate, n, kind="z2"):
        self.evaluate = evaluate
        self.n = n
        self.kind = kind
        self._entry = None

    def begin(self, entry_stream):
        self._entry = entry_stream

    def __call__(self, first_probe, count):
        g = self._entry.copy()
        g.jump(first_probe * self.n)
        return self.evaluate(g.probes(count, self.n, self.kind))


class DeviceProbes:
    """Probe source of the GPU estimators: the engines generate their probes in HBM at the
    probes' stream positions (k_mt_generate) and evaluate them; nothing of size n touches the host.
    The probes of the round expected next (`round_stride` probes further down the stream, set by the
    loop) are drawn on the engines' generation streams while the current round is being solved."""

    def __init__(self, mg_solver, params, method, level, kind="z2"):
        self.mg_solver = mg_solver
        self.params = params
        self.method = method
        self.level = level
        self.kind = kind
        self.round_stride = 0          # distance (in probes) to this rank's slice of the next round
        self._ready = None

    def begin(self, entry_stream):
        window = entry_stream.window()
        for eng in _engines(self.mg_solver):
            eng.stream_set(window)
        self._ready = None

    def __call__(self, first_probe, count):
        nxt = (first_probe + self.round_stride, count) if self.round_stride > 0 else None
        e, f, c, self._ready = probe_batch_generated(self.mg_solver, self.params, self.method, self.level,
                                                     first_probe, count, self.kind, prefetch=nxt,
                                                     ready=self._ready)
        return e, f, c


def run_probe_loop(evaluate, n, level_tol, max_nr_ests, batch, comm=None, min_index=5,
                   verbose=False, probe_type="z2"):
    """The probe loop of stoch_trace.py:137-154 / 386-406 in rounds of batched probes.

    `evaluate` is a probe source -- called as source(first_probe, count) for the probes
    [first_probe, first_probe + count) of the loop, counted from the position of the global NumPy
    stream on entry (:class:`DeviceProbes`, :class:`HostProbes`) -- or a plain callable
    evaluate(probes[int8, (k, n)]) -> (ests[k], iters_fine[k], iters_coarse[k]), which is wrapped
    in :class:`HostProbes`.  Every rank evaluates only its contiguous slice of a round.
    Returns dict(index, avg, dev, ests, iters_fine, iters_coarse, rounds) where `index` is the
    loop index at which the reference would have left the loop.  On return the global NumPy
    stream sits exactly where the one-by-one loop would have left it."""
    comm = comm or _dist.default_comm()
    source = evaluate if hasattr(evaluate, "begin") else HostProbes(evaluate, n, probe_type)
    entry = ProbeStream.from_numpy_state()
    source.begin(entry)
    ests = np.zeros(0, dtype=np.complex128)
    it_f = np.zeros(0, dtype=np.int64)
    it_c = np.zeros(0, dtype=np.int64)
    rounds = 0
    stop_index = None
    avg = dev = 0.0
    while ests.size < max_nr_ests:
        round_size = min(batch * comm.world, max_nr_ests - ests.size)
        first_new = ests.size
        lo, hi = comm.my_slice(round_size)
        if hasattr(source, "round_stride"):
            # the next round (if the loop goes on and is a full one) starts round_size probes further on
            source.round_stride = round_size if ests.size + 2 * round_size <= max_nr_ests else 0
        if hi > lo:
            e, f, c = source(first_new + lo, hi - lo)
        else:
            e, f, c = np.zeros(0, complex), np.zeros(0, np.int64), np.zeros(0, np.int64)
        e, f, c = comm.allgather_probe_results(e, f, c, round_size)
        ests = np.concatenate([ests, e])
        it_f = np.concatenate([it_f, f])
        it_c = np.concatenate([it_c, c])
        rounds += 1
        if verbose:
            # the reference's per-probe debug prints (stoch_trace.py:150-152): its own O(N^2) loop
            for i in range(first_new, ests.size):
                avg, dev = _stats(ests[:i + 1])
                err = dev / sqrt(i + 1)
                print(dev)
                print(err)
                print(level_tol)
                if i >= min_index and err < level_tol:
                    stop_index = i
                    break
        else:
            hit = first_stop_index(ests, first_new, level_tol, min_index)
            if hit is not None:
                stop_index, avg, dev = hit
        if stop_index is not None:
            break
    if stop_index is None:
        stop_index = ests.size - 1
        avg, dev = _stats(ests)
    k = stop_index + 1
    # leave the global stream where the one-by-one loop leaves it: k probes of n draws each
    entry.jump(k * n)
    np.random.set_state(entry.numpy_state())
    return {"index": stop_index, "avg": avg, "dev": dev, "ests": ests[:k],
            "iters_fine": it_f[:k], "iters_coarse": it_c[:k], "rounds": rounds,
            "solved": int(ests.size)}


def _setup_solver(A, params, announce=True, defer_coarse=False):
    mg_solver = MG(A)
    if defer_coarse and "defer_coarse_levels" not in params:
        # build-only: the flow uses level 0 only until its work model at the very end (MG.finish_setup)
        params = dict(params, defer_coarse_levels=True)
    mg_solver.coarsest_iters = 0
    mg_solver.coarsest_iters_tot = 0
    mg_solver.coarsest_iters_avg = 0
    mg_solver.nr_calls = 0
    print("MG setup phase ...", end='', flush=True)
    t0 = time.time()
    mg_solver.setup(dof=params['dof'], aggrs=params['aggrs'], max_levels=params['max_nr_levels'],
                    dim=2, acc_eigvs=params['accuracy_mg_eigvs'],
                    sys_type=params['problem_name'], params=params)
    print(" done. Time : " + str(time.time() - t0) + " seconds")
    print(mg_solver)
    deferred = getattr(mg_solver, "_pending", None) is not None
    nr_levels = mg_solver.total_levels if deferred else len(mg_solver.ml.levels)
    mg_solver.total_levels = nr_levels
    for i in range(nr_levels):
        mg_solver.coarsest_lev_iters[i] = 0
    if nr_levels < 3:
        raise Exception("Use three or more levels.")
    if not deferred:
        for i in range(nr_levels - 1):
            mg_solver.ml.levels[i].P = csr_matrix(mg_solver.ml.levels[i].P)
            mg_solver.ml.levels[i].R = csr_matrix(mg_solver.ml.levels[i].R)
    return mg_solver, nr_levels


def _rough_trace(mg_solver, params, n, Vx_rank, tr1):
    """stoch_trace.py:103-115 / 288-302: seed 123456, five deflated Hutchinson probes."""
    np.random.seed(123456)
    t0 = time.time()
    probes = draw_probes(NR_ROUGH_PROBES, n, params.get('probe_type', 'z2'))
    e, _, _ = probe_batch(mg_solver, params, "hutchinson", probes, 0)
    rough = np.sum(e[0:NR_ROUGH_PROBES]) / NR_ROUGH_PROBES
    rough += tr1
    print(" done. Time : " + str(time.time() - t0) + " seconds")
    return rough


# compute tr(A^{-1}) via (deflated) Hutchinson                      stoch_trace.py:33-179
def hutchinson(A, params):
    mg_solver, nr_levels = _setup_solver(A, params, defer_coarse=True)
    N = A.shape[0]
    batch = int(params.get('batch', DEFAULT_BATCH))

    print("\nResetting timer to zero ...", end='')
    mg_solver.timer.reset()
    print(" done\n")
    nr_deflat_vctrs = params['nr_deflat_vctrs']
    print("Computing deflation vectors ...", end='', flush=True)
    t0 = time.time()
    Vx, tr1 = deflation_pre_computations(A, nr_deflat_vctrs, params['defl_eigvs_tol_Hutch'],
                                         "hutchinson", mg_solver.timer, params, mg_solver)
    print(" done. Time : " + str(time.time() - t0) + " seconds")
    print(mg_solver.timer)

    print("\nComputing rough estimation of the trace ...", end='', flush=True)
    rough_trace = _rough_trace(mg_solver, params, N, Vx, tr1)
    rough_trace_tol = abs(params['tol'] * rough_trace)

    print("\nResetting timer to zero ...", end='')
    mg_solver.timer.reset()
    mg_solver.engine.timers_reset()
    print(" done")
    print("\nComputing the trace stochastically ...", end='', flush=True)
    t0 = time.time()
    mg_solver.coarsest_lev_iters[0] = 0

    source = DeviceProbes(mg_solver, params, "hutchinson", 0, params.get('probe_type', 'z2'))
    # a round = one batch per engine handle (concurrent HIP streams) per rank
    loop = run_probe_loop(source, N, rough_trace_tol, params['max_nr_ests'],
                          batch * max(1, len(_engines(mg_solver))),
                          verbose=bool(params.get('verbose', False)),
                          probe_type=params.get('probe_type', 'z2'))
    loop_s = time.time() - t0
    print(" done. Time : " + str(loop_s) + " seconds")

    function_iters = int(np.sum(loop["iters_fine"]))
    mg_solver.coarsest_lev_iters[0] = function_iters
    mg_solver.finish_setup()          # the coarse levels (built beside the probe loop): the work model reads their nnz
    result = dict()
    result['trace'] = loop["avg"] + tr1
    result['std_dev'] = loop["dev"]
    result['nr_ests'] = loop["index"]
    result['function_iters'] = function_iters
    levels = mg_solver.ml.levels
    result['total_complexity'] = flopsV_manual(len(levels), levels, 0, mg_solver) * function_iters
    result['total_complexity'] += levels[len(levels) - 1].A.nnz * mg_solver.coarsest_lev_iters[0]
    # stoch_trace.py:173-175 (hard-coded 1/3 kept)
    result['total_complexity'] += result['nr_ests'] * (2 * N * nr_deflat_vctrs) / 3.0
    # build-only extras (not in the reference's dictionary)
    result['ests'] = loop["ests"]
    result['rough_trace'] = rough_trace
    result['level_tol'] = rough_trace_tol
    result['probe_loop_s'] = loop_s                 # wall clock of the probe loop and what it solved
    result['probes_solved'] = loop["solved"]        # (whole rounds: >= nr_ests + 1)
    mg_solver.sync_timer()
    print(mg_solver.timer)
    return result


# compute tr(A^{-1}) via multigrid multilevel Monte Carlo          stoch_trace.py:185-471
def mlmc(A, params):
    skip_list = params['mlmc_levels_to_skip']
    if len(skip_list) > 1:
        raise Exception("Only allowed to skip one level for now")
    skip_level = len(skip_list) == 1
    if skip_level and not skip_list[0] == 1:
        raise Exception("Only allowed to skip the second level for now")

    mg_solver, nr_levels = _setup_solver(A, params)
    N = A.shape[0]
    batch = int(params.get('batch', DEFAULT_BATCH))
    mg_solver.skip_level = skip_level

    print("\nResetting timer to zero ...", end='')
    mg_solver.timer.reset()
    print(" done\n")
    print("Computing deflation vectors ...", end='', flush=True)
    t0 = time.time()
    nr_deflat_vctrs = params['mlmc_deflat_vctrs']
    tolx = params['defl_eigvs_tol_MLMC']
    tr1s = []
    for ix in range(nr_levels - 1):
        if skip_level and ix == 1:
            tr1s.append(0.0)
            continue
        # eigenvectors of the difference operator (A_f^-1 - P A_c^-1 R) g3   stoch_trace.py:257-270
        mg_solver.level_for_diff_op = ix
        n_ix = mg_solver.ml.levels[ix].A.shape[0]
        lop = LinearOperator((n_ix, n_ix), dtype=np.complex128,
                             matvec=lambda v: mg_solver.diff_op_Q(np.array(v, dtype=np.complex128)))
        _, _, tr1 = deflation_pre_computations(A, nr_deflat_vctrs[ix], tolx, "mlmc", mg_solver.timer,
                                               params, mg_solver, lop, level_nr=ix)
        tr1s.append(tr1)
    print(" done. Time : " + str(time.time() - t0) + " seconds")
    print(mg_solver.timer)

    print("Computing deflation vectors (for rough trace estimation purposes only) ...", end='',
          flush=True)
    t0 = time.time()
    Vx, tr1 = deflation_pre_computations(A, params['nr_deflat_vctrs'],
                                         params['defl_eigvs_tol_Hutch'], "hutchinson",
                                         mg_solver.timer, params, mg_solver)
    print(" done. Time : " + str(time.time() - t0) + " seconds")
    print("\nComputing rough estimation of the trace ...", end='', flush=True)
    rough_trace = _rough_trace(mg_solver, params, N, Vx, tr1)

    output_params = {'nr_levels': nr_levels, 'trace': 0.0, 'total_complexity': 0.0,
                     'std_dev': 0.0, 'results': [], 'rough_trace': rough_trace}
    for i in range(nr_levels):
        output_params['results'].append({'function_iters': 0, 'nr_ests': 0, 'ests_avg': 0.0,
                                         'ests_dev': 0.0, 'level_complexity': 0.0})

    # tolerance split between the difference levels               stoch_trace.py:327-336
    if nr_levels < 3:
        raise Exception("Number of levels restricted to >2 for now ...")
    if nr_levels == 3:
        frac0, frac1 = 0.8, 0.2
    else:
        frac0, frac1 = 0.45, 0.45
    if skip_level:
        frac0 = frac0 + frac1

    print("\nResetting timer to zero ...", end='')
    mg_solver.timer.reset()
    mg_solver.engine.timers_reset()
    print(" done\n")
    mg_solver.coarsest_lev_iters[0] = 0
    levels = mg_solver.ml.levels

    for i in range(nr_levels - 1):
        if skip_level and i == 1:
            continue
        t0 = time.time()
        if i == 0:
            tol_fctr = sqrt(frac0)
        elif i == 1:
            tol_fctr = sqrt(frac1)
        elif skip_level:
            tol_fctr = sqrt(1.0 - frac0) / sqrt(nr_levels - 3)
        else:
            tol_fctr = sqrt(1.0 - frac0 - frac1) / sqrt(nr_levels - 3)
        level_trace_tol = abs(params['tol'] * rough_trace * tol_fctr)
        n_i = levels[i].A.shape[0]
        lc = i + 2 if (skip_level and i == 0) else i + 1
        print("Computing for level " + str(i) + " ...", end='', flush=True)

        source = DeviceProbes(mg_solver, params, "mlmc", i, params.get('probe_type', 'z2'))
        loop = run_probe_loop(source, n_i, level_trace_tol, params['max_nr_ests'],
                              batch * max(1, len(_engines(mg_solver))),
                              probe_type=params.get('probe_type', 'z2'))
        res = output_params['results']
        res[i]['function_iters'] += int(np.sum(loop["iters_fine"]))
        res[lc]['function_iters'] += int(np.sum(loop["iters_coarse"]))
        mg_solver.coarsest_lev_iters[i] += int(np.sum(loop["iters_fine"]))
        res[i]['nr_ests'] += loop["index"]
        res[i]['ests_avg'] = loop["avg"] + tr1s[i]
        res[i]['ests_dev'] = loop["dev"]
        res[i]['ests'] = loop["ests"]              # build-only extras
        res[i]['level_tol'] = level_trace_tol
        res[i]['probe_loop_s'] = time.time() - t0
        res[i]['probes_solved'] = loop["solved"]
        print(" done. Time : " + str(time.time() - t0) + " seconds")

    # coarsest level, computed directly                            stoch_trace.py:418-437
    last = nr_levels - 1
    if levels[last].A.shape[0] == 1:
        raise Exception("your coarsest-level matrix is of size 1 ... is this what you want?")
    if params['coarsest_level_directly'] == True:   # noqa: E712  (as the reference tests it)
        output_params['results'][last]['nr_ests'] += 1
        crst_mat = mg_solver.coarsest_inv
        if params["use_permuted"]:
            crst_mat = levels[last].Pperm.transpose().conjugate() * (crst_mat * levels[last].Bblock_perm)
        output_params['results'][last]['ests_avg'] = np.trace(crst_mat)
        output_params['results'][last]['ests_dev'] = 0
    elif params.get('stochastic_coarsest'):
        # build-only option (the reference raises here): the coarsest term tr(Pperm^H A_c^-1 Bblock)
        # by plain Hutchinson probes on the coarsest level, with the tolerance share of the last
        # difference level
        n_c = levels[last].A.shape[0]
        if nr_levels == 3:
            tol_fctr = sqrt(1.0 - frac0) if skip_level else sqrt(frac1)
        elif skip_level:
            tol_fctr = sqrt(1.0 - frac0) / sqrt(nr_levels - 3)
        else:
            tol_fctr = sqrt(1.0 - frac0 - frac1) / sqrt(nr_levels - 3)
        level_trace_tol = abs(params['tol'] * rough_trace * tol_fctr)
        source = DeviceProbes(mg_solver, params, "level", last, params.get('probe_type', 'z2'))
        loop = run_probe_loop(source, n_c, level_trace_tol, params['max_nr_ests'], batch,
                              probe_type=params.get('probe_type', 'z2'))
        res = output_params['results'][last]
        res['function_iters'] += int(np.sum(loop["iters_fine"]))
        res['nr_ests'] += loop["index"]
        res['ests_avg'] = loop["avg"]
        res['ests_dev'] = loop["dev"]
        res['ests'] = loop["ests"]
        res['level_tol'] = level_trace_tol
    else:
        raise Exception("Stochastic coarsest-level computation is disabled at the moment.")

    # work model                                                    stoch_trace.py:443-467
    for i in range(nr_levels - 1):
        res = output_params['results'][i]
        res['level_complexity'] = res['function_iters'] * flopsV_manual(i, levels, i, mg_solver)
        res['level_complexity'] += levels[last].A.nnz * mg_solver.coarsest_lev_iters[i]
    nc = levels[last].A.shape[0]
    output_params['results'][last]['level_complexity'] = \
        pow(nc, 3) + output_params['results'][last]['function_iters'] * pow(nc, 2)
    for i in range(nr_levels):
        output_params['total_complexity'] += output_params['results'][i]['level_complexity']
        output_params['trace'] += output_params['results'][i]['ests_avg']
    mg_solver.sync_timer()
    print(mg_solver.timer)
    return output_params
