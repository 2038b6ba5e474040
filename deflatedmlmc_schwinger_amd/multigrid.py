"""``multigrid.MG`` of the reference (multigrid.py:56-557) on the MI355X engine.

Same class name, constructor, method names and attributes; the arithmetic
behind ``matvec`` / ``one_mg_step`` / ``solve`` / ``diff_op`` runs in the HIP library through
:mod:`deflatedmlmc_schwinger_amd.engine`.  Setup (ARPACK test vectors, Galerkin products,
dense inverse) stays on the host exactly as in the reference and is uploaded once.
"""
import os
import time

import numpy as np
import scipy.sparse as sp

from . import cache as _cache
from . import dist as _dist
from . import hierarchy as _hier
from .engine import Engine, EngineError
from .hierarchy import LevelML, SimpleML  # noqa: F401  (re-exported, multigrid.py:26-48)
from .utils import CustomTimer

REF_HID = 0      # reference hierarchy (MLMC level operators)
SOLVER_HID = 1   # level-0 preconditioner hierarchy


def collective_reference_hierarchy(A, dof, aggrs, max_levels, acc_eigvs, params, tv=None,
                                   comm=None):
    """hierarchy.reference_hierarchy for one or several ranks: with more than one rank, rank 0
    runs ARPACK and every rank builds the hierarchy from rank 0's test vectors (an iterative
    eigensolver is not guaranteed to be bit-reproducible across processes, and the MLMC level
    operators must be identical on all ranks)."""
    comm = comm or _dist.default_comm()
    built = None
    if tv is None and comm.world > 1:
        def root_build():
            nonlocal built
            built = _hier.reference_hierarchy(A, dof, aggrs, max_levels, acc_eigvs, params)
            return built[2]
        tv = comm.compute_on_root(root_build)
    if built is None:
        built = _hier.reference_hierarchy(A, dof, aggrs, max_levels, acc_eigvs, params,
                                          testvectors=tv)
    return built


def _new_engine(device):
    """An engine with the switches of SW_ENGINE_OPTS (comma-separated name=value, as bench.py's --engine-opts)
    set before anything is built: A/B runs of switches that act during the setup (e.g. gj_block)."""
    eng = Engine(device)
    for kv in [x for x in os.environ.get("SW_ENGINE_OPTS", "").split(",") if x]:
        eng.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    return eng


class MG:

    def __init__(self, A, smooth_iters=2):
        # level from which solves start (changes during MLMC)            multigrid.py:59-61
        self.level_nr = 0
        self.ml = []
        self.A = A
        self.x = []
        self.num_iters = 0
        self.total_levels = 0
        self.coarsest_iters = 0
        self.coarsest_iters_tot = 0
        self.coarsest_iters_avg = 0
        self.nr_calls = 0
        self.smooth_iters = smooth_iters
        self.coarsest_lev_iters = [0] * 10
        self.level_for_diff_op = 0
        self.solve_tol = 1.0e-1
        self._cinv = []           # (the attribute coarsest_inv of multigrid.py:84; a lazy property here)
        self.timer = CustomTimer()
        self.skip_level = False
        # build-specific state
        self.engine = None        # first engine handle (building blocks, single-RHS calls)
        self.engines = []         # all handles: concurrent probe batches use one stream each
        self.device = 0
        self.solver_info = None
        self.testvectors = None
        self._A0 = A
        self._have_solver_hier = False
        self.maxiter_cap = 1000
        self._ref_weights = {}
        self._lat = False           # cached hierarchy.detect_lattice(A) (False: not looked at yet)
        self._solver_cfg_built = None
        self._device_defl = {}      # (k, tol) -> (eigenvalues, eigenvectors) of gamma_3 A found during setup
        self.coarse_eo = None
        self.setup_log = {}

    # the dense inverse of the coarsest operator (multigrid.py:342-344).  With the device setup it is formed on
    # the GPU (sw_setup_invert_coarsest) and only comes to the host when somebody reads the attribute
    # (stoch_trace.py:428-435 traces it; plain Hutchinson flows never do).
    @property
    def coarsest_inv(self):
        if self._cinv is None and self.engine is not None and self.ml and len(self.ml.levels) > 1:
            n_c = self.ml.levels[-1].A.shape[0]
            M = self.engine.get_coarsest_inv(REF_HID, n_c)
            if self.coarse_eo is not None:
                # the engine holds the coarse dofs tile by tile (ref_coarsest = "eo"): back to the reference order
                pi = self.coarse_eo[0]
                out = np.empty_like(M)
                out[np.ix_(pi, pi)] = M
                M = out
            self._cinv = np.matrix(M)
        return self._cinv

    @coarsest_inv.setter
    def coarsest_inv(self, value):
        self._cinv = value

    def _lattice(self):
        if self._lat is False:
            self._lat = self._A0 if isinstance(self._A0, tuple) else _hier.detect_lattice(self._A0)
        return self._lat

    # ------------------------------------------------------------------------------------
    def setup(self, dof=[2, 8, 8], aggrs=[2 * 2, 2 * 2], max_levels=3, dim=2, acc_eigvs='low',
              sys_type='schwinger', params=None):
        """multigrid.py:100-345 -- the reference's index arithmetic on the host (aggregation, spin split,
        Gram-Schmidt, Galerkin products), its eigensolves on the GPU where the operator is a lattice operator
        (device_setup: level-0 test vectors and the deflation vectors by block subspace iteration with the
        engine's own multigrid solves as the shift-invert, the coarsest inverse by Gauss-Jordan on the device),
        on the host exactly as in the reference otherwise (params["setup_eigs"] = "reference")."""
        if params is None:
            raise Exception("setup needs the trace parameter dictionary")
        tv = params.get("mg_testvectors")
        cdir = _cache.cache_dir(params)
        ckey = None
        if cdir and tv is None:
            ckey = _cache.matrix_key(self._A0, {"dof": dof, "aggrs": aggrs, "levels": max_levels,
                                                "acc": acc_eigvs})
            hit = _cache.load(cdir, "mgtv", ckey)
            if hit is not None:
                tv = [hit["tv%d" % i] for i in range(max_levels - 1)]
                ckey = None
        self._cache_dir = cdir
        t0 = time.time()
        if self._device_setup_wanted(params) and params.get("defer_coarse_levels"):
            # plain / deflated Hutchinson flows never touch the coarse levels of the reference hierarchy on the
            # GPU: build them on a host thread WHILE the probes run and join at the end (finish_setup)
            self._deferred_reference_hierarchy(dof, aggrs, max_levels, acc_eigvs, params, tv, ckey)
            self.setup_log["reference_hierarchy_s"] = round(time.time() - t0, 4)
            return
        if self._device_setup_wanted(params):
            ml, cinv, used = self._device_reference_hierarchy(dof, aggrs, max_levels, acc_eigvs, params, tv)
        else:
            ml, cinv, used = collective_reference_hierarchy(self._A0, dof, aggrs, max_levels,
                                                            acc_eigvs, params, tv)
        self.setup_log["reference_hierarchy_s"] = round(time.time() - t0, 4)
        if ckey is not None:
            _cache.save(cdir, "mgtv", ckey, {"tv%d" % i: np.asarray(v) for i, v in enumerate(used)})
        self.ml = ml
        self.coarsest_inv = cinv
        self.testvectors = used
        self.total_levels = len(ml.levels)
        self.A = ml.levels[0].A
        t0 = time.time()
        self._upload(params)
        self.setup_log["upload_s"] = round(time.time() - t0, 4)

    # ---- device setup --------------------------------------------------------------------------
    def _solver_cfg_of(self, params):
        lat = self._lattice()
        cfg = params.get("solver_cfg") if params else None
        if (cfg is None or cfg == "auto") and lat is not None:
            cfg = _hier.auto_solver_cfg(lat[0])
        return cfg

    def _device_setup_wanted(self, params):
        """The device eigensolver serves this setup: a lattice operator, eigenvector test vectors (the preset's
        type), a solver hierarchy that is itself built on the device (its multigrid solves are the
        shift-invert operator) -- unless params["setup_eigs"] (or SW_SETUP_EIGS) says "reference"."""
        mode = params.get("setup_eigs", os.environ.get("SW_SETUP_EIGS", "device"))
        if mode != "device" or params.get("test_vectors_type") != "EVs":
            return False
        if self._lattice() is None:
            return False
        cfg = self._solver_cfg_of(params)
        return isinstance(cfg, dict) and cfg.get("setup") == "device"

    def _prepare_device(self, params):
        """Engines, the lattice operator and the level-0 solver hierarchy, ahead of the reference hierarchy."""
        device = int(params.get("device", self.device))
        nr_engines = max(1, int(params.get("engines", 1)))
        if self.engine is None:
            self.engines = [_new_engine(device) for _ in range(nr_engines)]
            self.engine = self.engines[0]
        lat = self._lattice()
        self.lattice = lat
        lev = LevelML()
        lev.A = self._A0 if not isinstance(self._A0, tuple) else None
        self.ml = SimpleML()
        self.ml.levels.append(lev)
        for eng in self.engines:
            eng.hier_begin(REF_HID, 1)
            eng.set_lattice(REF_HID, lat[0], lat[1], lat[2], lat[3])
            eng.hier_end(REF_HID)
        cfg = self._solver_cfg_of(params)
        self.upload_solver_hierarchy(cfg, params.get("solver_testvectors"))
        self._solver_cfg_built = dict(cfg)

    def device_eigenpairs(self, k, tol, hermitian=False, log=None):
        """k eigenpairs nearest zero of A (level 0) -- or of gamma_3 A (hermitian) -- on the GPU, the level-0
        solver hierarchy's solves as the shift-invert (setup_gpu.device_eigenpairs)."""
        from . import setup_gpu
        if not self._have_solver_hier:
            raise EngineError("the device eigensolver needs the level-0 solver hierarchy")
        return setup_gpu.device_eigenpairs(self.engine, SOLVER_HID, 0, k, tol, hermitian_g3=hermitian, log=log)

    def _deferred_reference_hierarchy(self, dof, aggrs, max_levels, acc_eigvs, params, tv, ckey):
        """Device setup for flows that use level 0 only (stoch_trace.hutchinson): engines, solver hierarchy,
        level-0 test vectors and deflation vectors on the GPU as in _device_reference_hierarchy; the host part
        (P_l, A_{l+1}, the small levels' ARPACK) starts on a thread and is NOT waited for -- finish_setup() joins
        it when somebody needs the coarse levels (the flow's work model reads their nnz at the very end).  The
        engine's reference hierarchy keeps the lattice level only."""
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.time()
        self._prepare_device(params)
        self.setup_log["solver_hierarchy_s"] = round(time.time() - t0, 4)
        comm = _dist.default_comm()
        tolx = 1.0e-3 if acc_eigvs == "low" else 1.0e-9
        kd = int(params.get("nr_deflat_vctrs", 0) or 0)
        tol_d = params.get("defl_eigvs_tol_Hutch", 1.0e-9)
        want_defl = kd > 0 and params.get("deflation_eigenpairs") is None and kd <= 32
        if want_defl and self._cache_dir:
            dkey = _cache.matrix_key(self._A0, {"k": kd, "tol": tol_d})
            want_defl = _cache.load(self._cache_dir, "defl", dkey) is None

        def root_job():
            tv0 = None
            if tv is None:
                log = []
                t1 = time.time()
                _, tv0 = self.device_eigenpairs(int(dof[1] / 2), tolx, log=log)
                self.setup_log["eigs_level0"] = {"seconds": round(time.time() - t1, 4), "steps": log}
            return tv0
        tv0 = comm.compute_on_root(root_job)
        tvs = tv if tv is not None else [tv0]
        self._pending_pool = ThreadPoolExecutor(max_workers=1)
        self._pending = self._pending_pool.submit(_hier.reference_hierarchy, self._A0, dof, aggrs, max_levels,
                                                  acc_eigvs, params, tvs, None, False)
        self._pending_ckey = ckey
        if want_defl:
            dlog = []
            t1 = time.time()
            defl = comm.compute_on_root(lambda: self.device_eigenpairs(kd, tol_d, hermitian=True, log=dlog))
            self.setup_log["eigsh_deflation"] = {"seconds": round(time.time() - t1, 4), "steps": dlog}
            self._device_defl[(kd, float(tol_d))] = defl
        # level 0 as the estimators read it (multigrid.py:130-155), the rest arrives with finish_setup()
        lev = self.ml.levels[0]
        n = self._A0.shape[0]
        sign = np.ones(n)
        sign[n // 2:] = -1.0
        lev.g3 = sp.diags([sign], [0])
        if params["use_permuted"]:
            shift0 = params["latt_dims"][0] * 2 * params["x_displacement"]
            lev.perm_shift = shift0
            lev.Pperm = _hier.shift_operator(n, shift0)
            lev.Bblock_perm = sp.identity(n, dtype=np.complex128, format="csr")
            for eng in self.engines:
                eng.set_perm(0, int(shift0))
        self.total_levels = max_levels
        self.A = lev.A
        self._cinv = None
        if params.get("stop_factor") is not None:
            for eng in self.engines:
                eng.set_option("stop_factor", float(params["stop_factor"]))

    def finish_setup(self):
        """Join the host thread that builds the coarse levels of a deferred setup (no-op otherwise).  The levels
        become available as self.ml.levels for the work model; they are NOT uploaded to the GPU (flows that need
        them there call setup() without defer_coarse_levels)."""
        pending = getattr(self, "_pending", None)
        if pending is None:
            return
        t0 = time.time()
        ml, _, used = pending.result()
        self._pending = None
        self._pending_pool.shutdown(wait=False)
        self.setup_log["host_levels_wait_s"] = round(time.time() - t0, 4)
        if self._pending_ckey is not None:
            _cache.save(self._cache_dir, "mgtv", self._pending_ckey,
                        {"tv%d" % i: np.asarray(v) for i, v in enumerate(used)})
        lev0 = self.ml.levels[0]
        ml.levels[0].A = lev0.A          # (the same operator object the callers hold)
        self.ml = ml
        self.testvectors = used
        self.total_levels = len(ml.levels)
        self.A = ml.levels[0].A
        self._coarse_levels_on_host_only = True

    def _device_reference_hierarchy(self, dof, aggrs, max_levels, acc_eigvs, params, tv):
        """hierarchy.reference_hierarchy with the eigensolves of the lattice level on the GPU.  Rank 0 runs
        them (every rank holds the same deterministic solver hierarchy; one source keeps the MLMC level
        operators identical by construction) and, while its host thread builds P_0, A_1 and the small coarse
        levels (host ARPACK on 8192 / 2048 rows), the GPU already computes the deflation vectors the
        estimators will ask for next.  The coarsest inverse is left to the engine (_upload)."""
        t0 = time.time()
        self._prepare_device(params)
        self.setup_log["solver_hierarchy_s"] = round(time.time() - t0, 4)
        comm = _dist.default_comm()
        tolx = 1.0e-3 if acc_eigvs == "low" else 1.0e-9
        kd = int(params.get("nr_deflat_vctrs", 0) or 0)
        tol_d = params.get("defl_eigvs_tol_Hutch", 1.0e-9)
        want_defl = kd > 0 and params.get("deflation_eigenpairs") is None and kd <= 32
        if want_defl and self._cache_dir:
            dkey = _cache.matrix_key(self._A0, {"k": kd, "tol": tol_d})
            want_defl = _cache.load(self._cache_dir, "defl", dkey) is None
        built = None

        def root_job():
            nonlocal built
            from concurrent.futures import ThreadPoolExecutor
            tvs = tv
            if tvs is None:
                log = []
                t1 = time.time()
                _, tv0 = self.device_eigenpairs(int(dof[1] / 2), tolx, log=log)
                self.setup_log["eigs_level0"] = {"seconds": round(time.time() - t1, 4), "steps": log}
                tvs = [tv0]
            with ThreadPoolExecutor(max_workers=1) as pool:
                fut = pool.submit(_hier.reference_hierarchy, self._A0, dof, aggrs, max_levels, acc_eigvs,
                                  params, tvs, None, False)
                defl = None
                if want_defl:
                    dlog = []
                    t1 = time.time()
                    defl = self.device_eigenpairs(kd, tol_d, hermitian=True, log=dlog)
                    self.setup_log["eigsh_deflation"] = {"seconds": round(time.time() - t1, 4), "steps": dlog}
                t1 = time.time()
                built = fut.result()
                self.setup_log["host_levels_wait_s"] = round(time.time() - t1, 4)
            return built[2], defl

        tvs, defl = comm.compute_on_root(root_job)
        if built is None:
            built = _hier.reference_hierarchy(self._A0, dof, aggrs, max_levels, acc_eigvs, params,
                                              testvectors=tvs, invert=False)
        if defl is not None:
            self._device_defl[(kd, float(tol_d))] = defl
        return built

    def setup_solver_only(self, solver_cfg=None, device=0, engines=1):
        """Large synthetic lattices (BASELINE config 5): no reference (MLMC) hierarchy, only the
        level-0 operator and the solver hierarchy, built without host ARPACK / SuperLU when
        ``solver_cfg["setup"] == "adaptive"``.  Plain / deflated Hutchinson probes only."""
        lat = self._lattice()
        if lat is None:
            raise Exception("setup_solver_only needs a lattice operator")
        self.lattice = lat
        lev = LevelML()
        lev.A = self._A0 if not isinstance(self._A0, tuple) else None
        self.ml = SimpleML()
        self.ml.levels.append(lev)
        self.total_levels = 1
        self.engines = [_new_engine(device) for _ in range(max(1, engines))]
        self.engine = self.engines[0]
        for eng in self.engines:
            eng.hier_begin(REF_HID, 1)
            eng.set_lattice(REF_HID, lat[0], lat[1], lat[2], lat[3])
            eng.hier_end(REF_HID)
        self.upload_solver_hierarchy(solver_cfg)

    def attach_hierarchy(self, ml, coarsest_inv, params):
        """Use an already built reference hierarchy (e.g. from a cache) instead of setup()."""
        self.ml = ml
        self.coarsest_inv = coarsest_inv
        self.total_levels = len(ml.levels)
        self.A = ml.levels[0].A
        self._upload(params)

    def _upload(self, params):
        device = int(params.get("device", self.device)) if params else self.device
        nr_engines = max(1, int(params.get("engines", 1))) if params else 1
        if self.engine is None:
            self.engines = [_new_engine(device) for _ in range(nr_engines)]
            self.engine = self.engines[0]
        levels = self.ml.levels
        nlev = len(levels)
        lat = self._lattice()
        self.lattice = lat
        # None: the engine forms the inverse itself from the coarsest operator (sw_setup_invert_coarsest)
        n_c = levels[-1].A.shape[0]
        if self._cinv is None and (n_c % 16 or n_c > 8192):
            self._cinv = np.matrix(_hier.dense_inverse(levels[-1].A.toarray()))
        cinv = None if self._cinv is None else np.asarray(self._cinv)
        rhsmaps = {}
        for i in range(nlev):
            if params and params.get("use_permuted") and not isinstance(levels[i].Pperm, int):
                rhsmaps[i] = sp.csr_matrix(levels[i].Bblock_perm @ levels[i].Pperm.transpose())
        # Build-only key ref_coarsest = "eo" (2-level reference hierarchy on a lattice, plain / deflated
        # Hutchinson runs): the coarse level is solved exactly in even-odd form -- dense inverse of the Schur
        # complement of its 16-row tiles (hierarchy.reference_coarse_eo), a quarter of the dense flops of
        # coarsest_inv.  The engine then holds the coarse dofs tile by tile (P's columns, A_c and the dense
        # inverse permuted alike; host-side attributes keep the reference's order) and the MLMC operands
        # of level 1 (perm, rhs map) are not uploaded.
        self.coarse_eo = None
        if params and params.get("ref_coarsest") == "eo" and nlev == 2 and lat is not None:
            aggr_size = levels[0].A.shape[0] // (levels[0].P.shape[1] // int(params.get("ref_coarse_dofs", 8)))
            self.coarse_eo = _hier.reference_coarse_eo(levels[0].P, levels[1].A, lat[0], aggr_size)
        if self.coarse_eo is not None:
            pi = self.coarse_eo[0]
            up_P = {0: sp.csr_matrix(levels[0].P)[:, pi]}
            up_A = {1: sp.csr_matrix(levels[1].A)[pi][:, pi]}
            if cinv is not None:
                cinv = cinv[np.ix_(pi, pi)]
            rhsmaps = {i: c for i, c in rhsmaps.items() if i == 0}
        else:
            up_P, up_A = {}, {}
        for eng in self.engines:
            eng.hier_begin(REF_HID, nlev)
            if lat is not None:
                L, mass, U1, U2 = lat
                eng.set_lattice(REF_HID, L, mass, U1, U2)
            else:
                eng.set_csr(REF_HID, 0, levels[0].A)
            for i in range(nlev - 1):
                if i > 0:
                    eng.set_csr(REF_HID, i, levels[i].A)
                eng.set_transfer(REF_HID, i, up_P.get(i, levels[i].P))
                # MR(nu) stands in for lgmres(maxiter=smooth_iters); see DESIGN.md section 4
                nu_post = int(params.get("ref_cycle_post", 4)) if params else 4
                eng.set_cycle(REF_HID, i, 0, nu_post,
                              int(params.get("ref_cycle_k", 0)) if params else 0)
                if params and params.get("ref_smoother") == "gmres30x2":
                    # reference-faithful cycle: lgmres(maxiter=smooth_iters), inner_m = 30
                    eng.set_gmres_smoother(REF_HID, i, 30, self.smooth_iters)
                if params and params.get("ref_smoother") == "richardson":
                    # fixed polynomial instead of MR(nu) on the reference hierarchy's levels
                    if ("ref", i, nu_post) not in self._ref_weights:
                        self._ref_weights[("ref", i, nu_post)] = _hier.smoother_weights(levels[i].A,
                                                                                        nu_post)
                    eng.set_smoother(REF_HID, i, None, self._ref_weights[("ref", i, nu_post)])
            if nlev > 1:
                eng.set_csr(REF_HID, nlev - 1, up_A.get(nlev - 1, levels[nlev - 1].A))
            if cinv is not None:
                eng.set_coarsest_inv(REF_HID, cinv)
            else:
                eng.setup_invert_coarsest(REF_HID)       # Gauss-Jordan on the GPU (multigrid.py:342-344)
            eng.hier_end(REF_HID)
            if self.coarse_eo is not None:
                for which, (tmap, kcol, vals) in enumerate(self.coarse_eo[1]):
                    eng.set_eo_operator(REF_HID, 1, which, tmap, kcol, vals)
                eng.setup_direct_level(REF_HID, 1)      # dense inverse of the Schur complement, on the device
            if params and params.get("ref_smoother") == "eo" and nlev > 1 and lat is not None:
                nu_post = int(params.get("ref_cycle_post", 4))
                # even-odd Schur-complement polynomial on the lattice level of the REFERENCE hierarchy
                # (half vectors, S four times better conditioned than A; outer solves then run on the
                # even-odd reduced system): weights from a device-side Arnoldi run on S
                key = ("ref-eo", nu_post)
                if key not in self._ref_weights:
                    if nu_post <= 32:
                        self._ref_weights[key] = _hier.weights_from_hessenberg(
                            eng.setup_arnoldi(REF_HID, 0, 1, nu_post))
                    else:
                        # beyond the device Arnoldi's 32 vectors: the same fit through the C ABI
                        from . import setup_gpu as _sg
                        self._ref_weights[key] = _hier.smoother_weights(
                            _sg._EngineSchur(eng, REF_HID, lat[0], lat[1]), nu_post)
                eng.set_eo_smoother(REF_HID, 0, self._ref_weights[key])
            # small intermediate levels of the reference hierarchy are solved directly (dense inverse formed
            # on the device): the MLMC coarse solves and coarse difference levels start there.  Not in the
            # reference-faithful mode, whose iteration counts must be the reference's.
            direct_max = int(params.get("ref_direct_max_n", 4096)) if params else 4096
            if not (params and params.get("ref_smoother") == "gmres30x2"):
                for i in range(1, nlev - 1):
                    n_i = levels[i].A.shape[0]
                    if n_i <= direct_max and n_i % 16 == 0:
                        eng.setup_level_inverse(REF_HID, i)
            for i, Cmat in rhsmaps.items():
                eng.set_perm(i, int(levels[i].perm_shift))
                eng.set_rhsmap(i, Cmat)
        if params and params.get("stop_factor") is not None:
            # build-only key: iterate every batch until its TRUE residuals are below stop_factor * tol
            # (iteration counts are still reported at tol, multigrid.py:347-366); 1 = the reference's point
            for eng in self.engines:
                eng.set_option("stop_factor", float(params["stop_factor"]))
        # level-0 preconditioner
        cfg = params.get("solver_cfg") if params else None
        if (cfg is None or cfg == "auto") and lat is not None:
            cfg = _hier.auto_solver_cfg(lat[0])      # tuned, device-built where the lattice allows it
        want = True if params is None else params.get("use_solver_hierarchy", True)
        if want and lat is not None:
            if not (self._have_solver_hier and self._solver_cfg_built == cfg):
                # (the device setup has built it already, ahead of the reference hierarchy)
                self._have_solver_hier = False
                self.upload_solver_hierarchy(cfg, params.get("solver_testvectors") if params else None)
                self._solver_cfg_built = dict(cfg) if isinstance(cfg, dict) else None
        else:
            # a solver hierarchy the device setup built for its eigensolves stays in its slot, unused
            self._have_solver_hier = False
            for eng in self.engines:
                eng.set_solver(int(params.get("solver_restart", 24)) if params else 24, REF_HID)

    def upload_solver_hierarchy(self, cfg=None, testvectors=None):
        """(Re)build the level-0 preconditioner hierarchy from `cfg` and make it the solver."""
        self._need_engine()
        lat = self.lattice
        if lat is None:
            raise Exception("the solver hierarchy needs a lattice operator at level 0")
        cfg = dict(_hier.DEFAULT_SOLVER_CFG if cfg is None else cfg)
        L = lat[0]
        Lf = L
        for (agg, _) in cfg["coarsening"]:
            if agg < 1 or Lf % agg:
                raise Exception("lattice extent %d not divisible by aggregate edge %d" % (Lf, agg))
            Lf //= agg
        t0 = time.time()
        cdir = getattr(self, "_cache_dir", None)
        skey = None
        if cdir and testvectors is None and self.ml.levels[0].A is not None:
            skey = _cache.matrix_key(self.ml.levels[0].A, {"coarsening": cfg["coarsening"],
                                                           "setup": cfg.get("setup", "eigs"),
                                                           "eig_tol": cfg.get("eig_tol")})
            hit = _cache.load(cdir, "solvertv", skey)
            if hit is not None:
                testvectors = [hit["tv%d" % i] for i in range(len(cfg["coarsening"]))]
                skey = None
        if cfg.get("setup", "eigs") == "device":
            # everything on the GPU, per engine (deterministic kernels: identical on every handle)
            from . import setup_gpu
            info = None
            for eng in self.engines:
                info = setup_gpu.device_solver_hierarchy(eng, lat, cfg, SOLVER_HID)
                eng.set_solver(int(cfg.get("restart", 24)), SOLVER_HID)
                eng.set_option("precond_f32", 1 if cfg.get("precond_precision", "f64") == "f32" else 0)
            self._have_solver_hier = True
            self.solver_hier = None
            self.solver_testvectors = None
            self.solver_info = {"levels": info["levels"], "setup_s": time.time() - t0, "cfg": cfg,
                                "setup_log": info["setup_log"]}
            return
        comm = _dist.default_comm()
        A0 = self.ml.levels[0].A
        if A0 is None:
            A0 = _hier.wilson_from_links(lat[2], lat[3], lat[0]) + \
                lat[1] * sp.identity(2 * lat[0] * lat[0], dtype=np.complex128, format="csr")
        adaptive = cfg.get("setup", "eigs") == "adaptive"

        def build(tvs):
            if adaptive and tvs is None:
                from . import setup_gpu
                return setup_gpu.adaptive_solver_hierarchy(self.engines[0], A0, lat, cfg, SOLVER_HID)
            return _hier.solver_hierarchy(A0, L, cfg, testvectors=tvs)

        sh = None
        if testvectors is None and comm.world > 1:
            # test vectors from ONE rank (ARPACK / the adaptive GPU setup), identical everywhere
            def root_build():
                nonlocal sh
                sh = build(None)
                return sh["tv"]
            testvectors = comm.compute_on_root(root_build)
        if sh is None:
            sh = build(testvectors)
        nl = len(sh["A"])
        if skey is not None:
            _cache.save(cdir, "solvertv", skey, {"tv%d" % i: np.asarray(v) for i, v in enumerate(sh["tv"])})
        self.solver_hier = sh
        self.solver_weights = []
        for i in range(nl - 1):
            cyc = cfg["cycle"][i]
            if cfg.get("smoother", "richardson") == "richardson":
                self.solver_weights.append((_hier.smoother_weights(sh["A"][i], cyc[0]),
                                            _hier.smoother_weights(sh["A"][i], cyc[1])))
            else:
                self.solver_weights.append(None)
        self.solver_weights_eo = None
        self.solver_eo = {}                   # level -> (weights, E rows, O rows) for the tests' model
        eo_levels = _hier.eo_levels_of(cfg) if cfg.get("smoother", "richardson") == "richardson" else []
        if 0 in eo_levels:
            S, E0, O0, _ = _hier.schur_complement(sh["A"][0], L)
            self.solver_weights_eo = _hier.smoother_weights(S, cfg["cycle"][0][1])
            self.solver_eo[0] = (self.solver_weights_eo, E0, O0)
        for eng in self.engines:
            eng.hier_begin(SOLVER_HID, nl)
            eng.set_lattice(SOLVER_HID, L, lat[1], lat[2], lat[3])
            for i in range(nl - 1):
                if i > 0:
                    eng.set_csr(SOLVER_HID, i, sh["A"][i])
                eng.set_transfer(SOLVER_HID, i, sh["P"][i])
                cyc = cfg["cycle"][i]
                eng.set_cycle(SOLVER_HID, i, cyc[0], cyc[1], cyc[2])
                if self.solver_weights[i] is not None:
                    eng.set_smoother(SOLVER_HID, i, self.solver_weights[i][0],
                                     self.solver_weights[i][1])
                if i == 0 and self.solver_weights_eo is not None:
                    eng.set_eo_smoother(SOLVER_HID, 0, self.solver_weights_eo)
            eng.set_coarsest_inv(SOLVER_HID, sh["coarsest_inv"])
            eng.hier_end(SOLVER_HID)
            eng.set_solver(int(cfg.get("restart", 24)), SOLVER_HID)
            eng.set_option("precond_f32", 1 if cfg.get("precond_precision", "f64") == "f32" else 0)
        Lc = L
        for i in range(1, nl - 1):
            Lc //= cfg["coarsening"][i - 1][0]
            if i in eo_levels:
                w, ops = _hier.upload_coarse_eo(self.engines, SOLVER_HID, i, sh["A"][i], Lc,
                                                cfg["cycle"][i][1])
                self.solver_eo[i] = (w, ops["E_rows"], ops["O_rows"])
        self._have_solver_hier = True
        self.solver_testvectors = sh["tv"]
        self.solver_info = {"levels": [a.shape[0] for a in sh["A"]],
                            "setup_s": time.time() - t0, "cfg": cfg,
                            "setup_log": sh.get("setup_log")}

    # ------------------------------------------------------------------------------------
    def _need_engine(self):
        if self.engine is None:
            raise EngineError("MG.setup() has not been run: no GPU engine attached")
        return self.engine

    def _level_of(self, A):
        for i, lev in enumerate(self.ml.levels):
            if lev.A is A:
                return i
        for i, lev in enumerate(self.ml.levels):
            if lev.A.shape == A.shape:
                return i
        raise Exception("matrix does not belong to the multigrid hierarchy")

    def _hid_for(self, level):
        return SOLVER_HID if (level == 0 and self._have_solver_hier) else REF_HID

    def solve(self, A, b, tol):
        """multigrid.py:347-366: result in self.x, iteration count in self.num_iters."""
        eng = self._need_engine()
        n = A.shape[0]
        maxiter = n if n < 1000 else self.maxiter_cap
        lvl = self.level_nr
        self.A = self.ml.levels[lvl].A
        x, its, _ = eng.solve(self._hid_for(lvl), lvl, np.asarray(b).reshape(-1), tol, maxiter)
        self.x = x
        self.num_iters = its

    def solve_batch(self, level, B, tol, maxiter=None):
        """Multi-RHS form of solve(): B has one right-hand side per row."""
        eng = self._need_engine()
        n = self.ml.levels[level].A.shape[0]
        if maxiter is None:
            maxiter = n if n < 1000 else self.maxiter_cap
        return eng.solve(self._hid_for(level), level, B, tol, maxiter)

    def one_mg_step(self, b):
        """multigrid.py:369-447: one cycle starting at self.level_nr."""
        eng = self._need_engine()
        lvl = self.level_nr
        nlev = len(self.ml.levels)
        self.coarsest_lev_iters[lvl] += 1
        self.coarsest_iters = 1
        self.nr_calls += 1
        self.coarsest_iters_tot += 1
        self.coarsest_iters_avg = self.coarsest_iters_tot / self.nr_calls
        if lvl == nlev - 1:
            return self._coarse_out(eng.coarsest(REF_HID, self._coarse_in(np.asarray(b).reshape(-1), lvl)), lvl)
        return eng.vcycle(self._hid_for(lvl), lvl, np.asarray(b).reshape(-1))

    # ref_coarsest = "eo" (2-level hierarchies): the ENGINE holds the coarse dofs tile by tile, the host
    # attributes and every vector that crosses these methods keep the reference's order
    def _coarse_in(self, v, level):
        if self.coarse_eo is None or level != 1:
            return v
        return np.ascontiguousarray(np.asarray(v)[..., self.coarse_eo[0]])

    def _coarse_out(self, y, level):
        if self.coarse_eo is None or level != 1:
            return y
        out = np.empty_like(y)
        out[..., self.coarse_eo[0]] = y
        return out

    def matvec(self, x):
        """multigrid.py:552-557: y = self.A * x on the GPU."""
        eng = self._need_engine()
        lvl = self._level_of(self.A)
        return eng.apply_dirac(REF_HID, lvl, np.asarray(x).reshape(-1))

    def __str__(self):
        out = "\nMultilevel information:\n"
        if getattr(self, "_pending", None) is not None:
            return out + "Level: 0\n\tsize(A) = " + str(self.ml.levels[0].A.shape) + \
                "\n(coarse levels: under construction on a host thread, MG.finish_setup() joins it)\n"
        last = len(self.ml.levels) - 1
        for idx, level in enumerate(self.ml.levels):
            out += "Level: " + str(idx) + "\n"
            if idx < last:
                out += "\tsize(R) = " + str(level.R.shape) + "\n"
                out += "\tsize(P) = " + str(level.P.shape) + "\n"
            out += "\tsize(A) = " + str(level.A.shape) + "\n"
        return out

    # ------------------------------------------------------------------------------------
    def diff_op_Q(self, v):
        """multigrid.py:461-468.  Like the reference, the sign flip is applied IN PLACE to
        the caller's array (``vx = v[:]`` is a view there)."""
        half = int(v.shape[0] / 2)
        v[half:] = -v[half:]
        return self.diff_op(v)

    def diff_op(self, v):
        """multigrid.py:471-549: (A_f^-1 - P A_c^-1 R) v at self.level_for_diff_op."""
        eng = self._need_engine()
        lvl = self.level_for_diff_op
        nlev = len(self.ml.levels)
        skip = self.skip_level and lvl == 0
        v = np.asarray(v, dtype=np.complex128).reshape(-1)
        if self.coarse_eo is not None and lvl != 0:
            raise Exception("ref_coarsest = 'eo' keeps the coarse level in tile order on the GPU: the difference "
                            "operator is available at level 0 only in that mode")
        vc = eng.restrict(REF_HID, lvl, v)       # (with coarse_eo: tile order in, tile order out below)
        lc = lvl + 1
        if skip:
            vc = eng.restrict(REF_HID, lvl + 1, vc)
            lc = lvl + 2
        self.level_nr = lvl
        self.solve(self.ml.levels[lvl].A, v, self.solve_tol)
        t1 = self.x
        if lc == nlev - 1:
            t2 = eng.coarsest(REF_HID, vc)
        else:
            self.level_nr = lc
            self.solve(self.ml.levels[lc].A, vc, self.solve_tol)
            t2 = self.x
        if skip:
            t2 = eng.prolong(REF_HID, lvl + 1, t2)
        return t1 - eng.prolong(REF_HID, lvl, t2)

    # ------------------------------------------------------------------------------------
    def sync_timer(self):
        """Copy the engine's HIP-event buckets (ms) into the CustomTimer fields (s)."""
        if self.engine is None:
            return
        t = self.engine.timers()
        self.timer.mvm = (t["mvm"] + t["coarsest"]) * 1e-3
        self.timer.defl = t["defl"] * 1e-3
        self.timer.P = t["P"] * 1e-3
        self.timer.R = t["R"] * 1e-3
        self.timer.axpy = t["axpy"] * 1e-3
        self.timer.dots = t["dots"] * 1e-3
