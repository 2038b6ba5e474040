"""ctypes binding of the C ABI in ``include/schwinger_hip.h`` (libschwinger_hip.so).

The library is the product's only compute path.  If it is missing, or no HIP
device is visible, construction of :class:`Engine` raises -- there is no CPU
fallback in this package (the NumPy restatement under ``oracle/`` is test
infrastructure and is never imported from here).
"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libschwinger_hip.so")

MODE_HUTCHINSON = 0
MODE_MLMC = 1
MODE_MLMC_SKIP = 2
MODE_LEVEL = 3
PROBES_Z2 = 1
PROBES_Z4 = 2
PROBE_KINDS = {"z2": PROBES_Z2, "z4": PROBES_Z4}

TIMER_NAMES = ("mvm", "defl", "P", "R", "axpy", "dots", "coarsest", "other")


class EngineError(RuntimeError):
    pass


_lib = None


def load_library():
    """Load the C-ABI library (raises EngineError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "HIP engine library not built: %s is missing (run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `make -C deflatedmlmc_schwinger_amd/csrc`)"
            % LIB_PATH)
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 and "torch" not in sys.modules:
        # A multi-rank launch will use torch.distributed (backend nccl = RCCL).  PyTorch bundles its own
        # ROCm runtime: loaded AFTER this library it is mapped as a second HIP / HSA runtime in the process
        # (same sonames, different files: /proc/self/maps shows both), loaded BEFORE it the dynamic loader
        # resolves this library's libamdhip64.so.7 / libhsa-runtime64.so.1 to PyTorch's mapped copies --
        # one runtime serving both.  So: torch first.
        try:
            import torch  # noqa: F401
        except Exception:      # no torch: the engine runs on the system runtime alone
            pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise EngineError("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    P = C.POINTER

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("sw_create", i32, P(vp), i32)
    sig("sw_destroy", i32, vp)
    sig("sw_last_error", C.c_char_p, vp)
    sig("sw_device_count", i32)
    sig("sw_version", C.c_char_p)
    sig("sw_hier_begin", i32, vp, i32, i32)
    sig("sw_set_lattice", i32, vp, i32, i32, dbl, vp, vp)
    sig("sw_set_csr", i32, vp, i32, i32, i32, vp, vp, vp)
    sig("sw_set_transfer", i32, vp, i32, i32, i32, i32, vp, vp, vp)
    sig("sw_set_coarsest_inv", i32, vp, i32, i32, vp)
    sig("sw_set_cycle", i32, vp, i32, i32, i32, i32, i32)
    sig("sw_set_smoother", i32, vp, i32, i32, i32, vp, i32, vp)
    sig("sw_set_gmres_smoother", i32, vp, i32, i32, i32, i32)
    sig("sw_set_eo_smoother", i32, vp, i32, i32, i32, vp)
    sig("sw_set_eo_operator", i32, vp, i32, i32, i32, i32, i32, vp, vp, vp)
    sig("sw_setup_eo_operators", i32, vp, i32, i32, i32)
    sig("sw_apply_eo_operator", i32, vp, i32, i32, i32, i32, vp, vp)
    sig("sw_get_level_bsr", i32, vp, i32, i32, P(i32), vp, vp)
    sig("sw_setup_testvectors", i32, vp, i32, i32, i32, C.c_uint64, i32, dbl, i32, i32, vp)
    sig("sw_setup_transfer", i32, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp)
    sig("sw_setup_galerkin", i32, vp, i32, i32, i32, vp)
    sig("sw_get_level_dense", i32, vp, i32, i32, vp)
    sig("sw_setup_invert_coarsest", i32, vp, i32)
    sig("sw_setup_direct_level", i32, vp, i32, i32)
    sig("sw_setup_level_inverse", i32, vp, i32, i32)
    sig("sw_setup_arnoldi", i32, vp, i32, i32, i32, i32, C.c_uint64, vp)
    sig("sw_hier_end", i32, vp, i32)
    sig("sw_set_deflation", i32, vp, i32, vp)
    sig("sw_set_level_deflation", i32, vp, i32, i32, vp)
    sig("sw_set_perm", i32, vp, i32, i64)
    sig("sw_set_rhsmap", i32, vp, i32, i32, vp, vp, vp)
    sig("sw_set_solver", i32, vp, i32, i32)
    sig("sw_set_option", i32, vp, C.c_char_p, dbl)
    sig("sw_get_option", i32, vp, C.c_char_p, P(dbl))
    sig("sw_pool_trim", i32)
    sig("sw_get_coarsest_inv", i32, vp, i32, vp)
    sig("sw_eig_begin", i32, vp, i32, i32, C.c_uint64)
    sig("sw_eig_load", i32, vp, i32, i32, vp)
    sig("sw_eig_solve", i32, vp, i32, i32, i32, dbl, i32, P(i32))
    sig("sw_eig_gram", i32, vp, i32, i32, vp)
    sig("sw_eig_rotate", i32, vp, i32, vp, i32, i32)
    sig("sw_eig_fetch", i32, vp, i32, i32, vp)
    sig("sw_eig_end", i32, vp)
    sig("sw_apply_dirac", i32, vp, i32, i32, i32, vp, vp)
    sig("sw_restrict", i32, vp, i32, i32, i32, vp, vp)
    sig("sw_prolong", i32, vp, i32, i32, i32, vp, vp)
    sig("sw_coarsest", i32, vp, i32, i32, vp, vp)
    sig("sw_vcycle", i32, vp, i32, i32, i32, vp, vp)
    sig("sw_solve", i32, vp, i32, i32, i32, vp, vp, dbl, i32, vp, vp)
    sig("sw_hutch_batch", i32, vp, i32, i32, i32, vp, dbl, i32, vp, vp)
    sig("sw_probes_upload", i32, vp, i32, i32, vp)
    sig("sw_probes_upload_slot", i32, vp, i32, i32, i32, vp)
    sig("sw_probes_select", i32, vp, i32)
    sig("sw_kernel_stats", i32, vp, i32, P(dbl), P(i64))
    sig("sw_kernel_work", i32, vp, i32, P(dbl))
    sig("sw_comm_unique_id", i32, vp)
    sig("sw_comm_init", i32, vp, i32, i32, vp)
    sig("sw_allreduce_stats", i32, vp, vp)
    sig("sw_comm_destroy", i32, vp)
    sig("sw_hutch_run", i32, vp, i32, i32, dbl, i32)
    sig("sw_sync", i32, vp)
    sig("sw_hutch_fetch", i32, vp, vp, vp)
    sig("sw_bench_dirac", i32, vp, i32, i32, i32, i32, P(dbl))
    sig("sw_set_profiling", i32, vp, i32)
    sig("sw_timers", i32, vp, P(dbl))
    sig("sw_timers_reset", i32, vp)
    sig("sw_launch_count", i32, vp, P(i64))
    sig("sw_mt_create", vp, C.c_uint32)
    sig("sw_mt_destroy", None, vp)
    sig("sw_mt_skip", None, vp, C.c_uint64)
    sig("sw_mt_raw", None, vp, C.c_uint64, vp)
    sig("sw_mt_rademacher", None, vp, C.c_uint64, vp)
    sig("sw_mt_z4", None, vp, C.c_uint64, vp)
    sig("sw_mt_from_state", vp, vp, i32)
    sig("sw_mt_get_state", None, vp, vp, P(i32))
    sig("sw_mt_window", None, vp, vp)
    sig("sw_mt_jump", i32, vp, C.c_uint64)
    sig("sw_mt_jump_poly", i32, C.c_uint64, vp)
    sig("sw_mt_window_jump", i32, vp, C.c_uint64, vp)
    sig("sw_probes_stream_set", i32, vp, vp)
    sig("sw_probes_generate", i32, vp, i32, i32, i32, i32, C.c_uint64)
    sig("sw_probes_fetch", i32, vp, i32, vp)
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "sw_create", "sw_destroy", "sw_last_error", "sw_device_count", "sw_version", "sw_hier_begin",
    "sw_set_lattice", "sw_set_csr", "sw_set_transfer", "sw_set_coarsest_inv", "sw_set_cycle",
    "sw_set_smoother", "sw_set_gmres_smoother", "sw_set_eo_smoother", "sw_set_eo_operator", "sw_setup_eo_operators", "sw_apply_eo_operator",
    "sw_get_level_bsr", "sw_setup_testvectors", "sw_setup_transfer",
    "sw_setup_galerkin", "sw_get_level_dense", "sw_setup_invert_coarsest", "sw_setup_direct_level", "sw_setup_level_inverse", "sw_setup_arnoldi", "sw_hier_end", "sw_set_deflation", "sw_set_level_deflation", "sw_set_perm", "sw_set_rhsmap", "sw_set_solver", "sw_set_option", "sw_get_option",
    "sw_pool_trim", "sw_get_coarsest_inv", "sw_eig_begin", "sw_eig_load", "sw_eig_solve", "sw_eig_gram", "sw_eig_rotate", "sw_eig_fetch", "sw_eig_end",
    "sw_apply_dirac", "sw_restrict", "sw_prolong", "sw_coarsest", "sw_vcycle", "sw_solve",
    "sw_hutch_batch", "sw_probes_upload", "sw_probes_upload_slot", "sw_probes_select",
    "sw_kernel_stats", "sw_kernel_work", "sw_hutch_run", "sw_sync", "sw_hutch_fetch",
    "sw_comm_unique_id", "sw_comm_init", "sw_allreduce_stats", "sw_comm_destroy",
    "sw_bench_dirac", "sw_set_profiling", "sw_timers", "sw_timers_reset", "sw_launch_count",
    "sw_mt_create", "sw_mt_destroy", "sw_mt_skip", "sw_mt_raw", "sw_mt_rademacher",
    "sw_mt_z4", "sw_mt_from_state", "sw_mt_get_state", "sw_mt_window", "sw_mt_jump",
    "sw_mt_jump_poly", "sw_mt_window_jump",
    "sw_probes_stream_set", "sw_probes_generate", "sw_probes_fetch",
)


def pool_trim():
    """Release the device memory the engines' process-wide block pool has parked."""
    return load_library().sw_pool_trim()


def device_count():
    return int(load_library().sw_device_count())


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _csr_parts(M):
    M = sp.csr_matrix(M)
    M.sort_indices()
    indptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(M.indices, dtype=np.int32)
    data = _c128(M.data)
    return M.shape, indptr, indices, data


class Engine:
    """One handle = one GPU + one stream.  Vectors passed in and out are NumPy arrays in the
    reference ordering, shape (n,) or (nb, n) (one row per right-hand side)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.sw_create(C.byref(self._h), int(device))
        if rc != 0:
            msg = self._lib.sw_last_error(None)
            self._h = None
            raise EngineError("sw_create failed: %s" % (msg.decode() if msg else "unknown"))
        self.device = int(device)
        self.level_sizes = {}

    # -- plumbing --------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc != 0:
            msg = self._lib.sw_last_error(self._h)
            raise EngineError("%s failed: %s" % (what, msg.decode() if msg else "unknown"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sw_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- operands --------------------------------------------------------------------------
    def hier_begin(self, hid, nlevels):
        self._chk(self._lib.sw_hier_begin(self._h, hid, nlevels), "sw_hier_begin")
        self.level_sizes[hid] = [0] * nlevels

    def set_lattice(self, hid, L, mass, U1, U2):
        U1, U2 = _c128(U1), _c128(U2)
        if U1.size != L * L or U2.size != L * L:
            raise EngineError("link arrays must have L*L entries")
        self._chk(self._lib.sw_set_lattice(self._h, hid, L, float(mass), _ptr(U1), _ptr(U2)),
                  "sw_set_lattice")
        self.level_sizes[hid][0] = 2 * L * L

    def set_csr(self, hid, level, A):
        (n, m), indptr, indices, data = _csr_parts(A)
        if n != m:
            raise EngineError("level operator must be square")
        self._chk(self._lib.sw_set_csr(self._h, hid, level, n, _ptr(indptr), _ptr(indices),
                                       _ptr(data)), "sw_set_csr")
        self.level_sizes[hid][level] = n

    def set_transfer(self, hid, level, P):
        (nf, nc), indptr, indices, data = _csr_parts(P)
        self._chk(self._lib.sw_set_transfer(self._h, hid, level, nf, nc, _ptr(indptr),
                                            _ptr(indices), _ptr(data)), "sw_set_transfer")
        self.level_sizes[hid][level] = nf
        self.level_sizes[hid][level + 1] = nc

    def set_coarsest_inv(self, hid, M):
        M = _c128(np.asarray(M))
        if M.ndim != 2 or M.shape[0] != M.shape[1]:
            raise EngineError("coarsest inverse must be a square dense matrix")
        self._chk(self._lib.sw_set_coarsest_inv(self._h, hid, M.shape[0], _ptr(M)),
                  "sw_set_coarsest_inv")
        self.level_sizes[hid][-1] = M.shape[0]

    def set_cycle(self, hid, level, nu_pre, nu_post, kcycle=0):
        self._chk(self._lib.sw_set_cycle(self._h, hid, level, nu_pre, nu_post, kcycle),
                  "sw_set_cycle")

    def set_smoother(self, hid, level, w_pre, w_post):
        wp = _c128(np.asarray(w_pre if w_pre is not None else [], dtype=np.complex128))
        wq = _c128(np.asarray(w_post if w_post is not None else [], dtype=np.complex128))
        self._chk(self._lib.sw_set_smoother(self._h, hid, level, wp.size,
                                            _ptr(wp) if wp.size else None, wq.size,
                                            _ptr(wq) if wq.size else None), "sw_set_smoother")

    def set_eo_smoother(self, hid, level, w_post):
        wq = _c128(np.asarray(w_post if w_post is not None else [], dtype=np.complex128))
        self._chk(self._lib.sw_set_eo_smoother(self._h, hid, level, wq.size,
                                               _ptr(wq) if wq.size else None), "sw_set_eo_smoother")

    def set_eo_operator(self, hid, level, which, tmap, kcol, vals):
        tmap = np.ascontiguousarray(tmap, dtype=np.int32)
        kcol = np.ascontiguousarray(kcol, dtype=np.int32)
        vals = _c128(vals)
        RT, KS = kcol.shape
        if vals.shape != (RT, KS, 64) or tmap.shape != (RT,):
            raise EngineError("even-odd operator arrays have inconsistent shapes")
        self._chk(self._lib.sw_set_eo_operator(self._h, hid, level, int(which), RT, KS, _ptr(tmap),
                                               _ptr(kcol), _ptr(vals)), "sw_set_eo_operator")

    def setup_eo_operators(self, hid, level, Lc):
        """S, F, G, Hb of block level `level` (Lc x Lc sites) built on the device from its operator."""
        self._chk(self._lib.sw_setup_eo_operators(self._h, hid, level, int(Lc)), "sw_setup_eo_operators")

    def apply_eo_operator(self, hid, level, which, X):
        X2, single = self._io(X, self._n(hid, level))
        Y = np.empty_like(X2)
        self._chk(self._lib.sw_apply_eo_operator(self._h, hid, level, int(which), X2.shape[0], _ptr(X2),
                                                 _ptr(Y)), "sw_apply_eo_operator")
        return Y[0] if single else Y

    def level_bsr(self, hid, level):
        """A (device-built) level operator in MFMA block-row form: (kcol[RT, KS], vals[RT, KS, 64])."""
        ks = C.c_int(0)
        self._chk(self._lib.sw_get_level_bsr(self._h, hid, level, C.byref(ks), None, None),
                  "sw_get_level_bsr")
        RT = self.level_sizes[hid][level] // 16
        kcol = np.empty((RT, ks.value), dtype=np.int32)
        vals = np.empty((RT, ks.value, 64), dtype=np.complex128)
        self._chk(self._lib.sw_get_level_bsr(self._h, hid, level, C.byref(ks), _ptr(kcol), _ptr(vals)),
                  "sw_get_level_bsr")
        return kcol, vals

    def set_gmres_smoother(self, hid, level, m, cycles):
        self._chk(self._lib.sw_set_gmres_smoother(self._h, hid, level, int(m), int(cycles)),
                  "sw_set_gmres_smoother")

    # -- GPU-side setup ---------------------------------------------------------------------
    def setup_testvectors(self, hid, level, nvec, seed, sweeps, tol, maxiter, precond):
        its = np.zeros(max(1, sweeps), dtype=np.int32)
        self._chk(self._lib.sw_setup_testvectors(self._h, hid, level, nvec, int(seed), sweeps,
                                                 float(tol), int(maxiter), 1 if precond else 0,
                                                 _ptr(its)), "sw_setup_testvectors")
        return its[:sweeps].tolist()

    def setup_transfer(self, hid, level, blk_rows, G, pcols, pmap, porder=None):
        blk_rows = np.ascontiguousarray(blk_rows, dtype=np.int32)
        pcols = np.ascontiguousarray(pcols, dtype=np.int32)
        pmap = np.ascontiguousarray(pmap, dtype=np.int64)
        nblocks, rpb = blk_rows.shape
        K = pcols.shape[1]
        if porder is not None:
            porder = np.ascontiguousarray(porder, dtype=np.int32)
        self._chk(self._lib.sw_setup_transfer(self._h, hid, level, nblocks, rpb, _ptr(blk_rows), G, K,
                                              _ptr(pcols), _ptr(pmap),
                                              _ptr(porder) if porder is not None else None),
                  "sw_setup_transfer")
        self.level_sizes[hid][level] = nblocks * rpb
        self.level_sizes[hid][level + 1] = nblocks * 8

    def setup_galerkin(self, hid, level, Lc, nbr):
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        self._chk(self._lib.sw_setup_galerkin(self._h, hid, level, int(Lc), _ptr(nbr)),
                  "sw_setup_galerkin")

    def setup_invert_coarsest(self, hid):
        self._chk(self._lib.sw_setup_invert_coarsest(self._h, hid), "sw_setup_invert_coarsest")
        self.level_sizes[hid][-1] = self.level_sizes[hid][-1]

    def get_coarsest_inv(self, hid, n):
        """The dense coarsest inverse the engine holds, as an (n, n) complex128 array."""
        out = np.empty((n, n), dtype=np.complex128)
        self._chk(self._lib.sw_get_coarsest_inv(self._h, hid, _ptr(out)), "sw_get_coarsest_inv")
        return out

    def setup_direct_level(self, hid, level):
        """Dense inverse of block level `level`'s even-odd Schur complement, formed on the device and
        installed as the level's even-odd operator 4 (the level is then solved exactly)."""
        self._chk(self._lib.sw_setup_direct_level(self._h, hid, level), "sw_setup_direct_level")

    def setup_level_inverse(self, hid, level):
        """Dense inverse of a small level's operator (n <= 8192), formed on the device; solves that start
        at this level are then two applications of it around one residual instead of an iteration."""
        self._chk(self._lib.sw_setup_level_inverse(self._h, hid, level), "sw_setup_level_inverse")

    def setup_arnoldi(self, hid, level, which, degree, seed=2024):
        """(degree+1) x degree Hessenberg matrix of `degree` Arnoldi steps of the level operator
        (which = 0) or its even-odd Schur complement (which = 1), computed on the device."""
        Hm = np.zeros((degree + 1, degree), dtype=np.complex128)
        self._chk(self._lib.sw_setup_arnoldi(self._h, hid, level, int(which), int(degree), int(seed),
                                             _ptr(Hm)), "sw_setup_arnoldi")
        return Hm

    def level_dense(self, hid, level):
        n = self.level_sizes[hid][level]
        M = np.empty((n, n), dtype=np.complex128)
        self._chk(self._lib.sw_get_level_dense(self._h, hid, level, _ptr(M)), "sw_get_level_dense")
        return M

    def hier_end(self, hid):
        self._chk(self._lib.sw_hier_end(self._h, hid), "sw_hier_end")

    def set_deflation(self, U):
        if U is None:
            self._chk(self._lib.sw_set_deflation(self._h, 0, None), "sw_set_deflation")
            return
        U = _c128(np.asarray(U))
        self._chk(self._lib.sw_set_deflation(self._h, U.shape[1], _ptr(U)), "sw_set_deflation")

    def set_level_deflation(self, level, V):
        if V is None:
            self._chk(self._lib.sw_set_level_deflation(self._h, level, 0, None),
                      "sw_set_level_deflation")
            return
        V = _c128(np.asarray(V))
        self._chk(self._lib.sw_set_level_deflation(self._h, level, V.shape[1], _ptr(V)),
                  "sw_set_level_deflation")

    def set_perm(self, level, shift):
        self._chk(self._lib.sw_set_perm(self._h, level, int(shift)), "sw_set_perm")

    def set_rhsmap(self, level, Cmat):
        (n, m), indptr, indices, data = _csr_parts(Cmat)
        self._chk(self._lib.sw_set_rhsmap(self._h, level, n, _ptr(indptr), _ptr(indices),
                                          _ptr(data)), "sw_set_rhsmap")

    def set_solver(self, restart=24, solver_hid=0):
        self._chk(self._lib.sw_set_solver(self._h, restart, solver_hid), "sw_set_solver")

    def set_option(self, name, value):
        self._chk(self._lib.sw_set_option(self._h, name.encode(), float(value)), "sw_set_option")

    def get_option(self, name):
        """Current value of an engine switch (save before an A/B run, restore afterwards)."""
        v = C.c_double(0.0)
        self._chk(self._lib.sw_get_option(self._h, name.encode(), C.byref(v)), "sw_get_option")
        return float(v.value)

    # -- device eigensolver (block subspace iteration; driver: setup_gpu.device_eigenpairs) ------
    def eig_begin(self, hid, level, seed=11):
        self._chk(self._lib.sw_eig_begin(self._h, hid, level, int(seed)), "sw_eig_begin")
        self._eig_n = self._n(hid, level)

    def eig_load(self, dst, X):
        X = np.ascontiguousarray(np.atleast_2d(X), dtype=np.complex128)
        self._chk(self._lib.sw_eig_load(self._h, dst, X.shape[0], _ptr(X)), "sw_eig_load")

    def eig_solve(self, src, dst, mode, tol, maxiter=1000):
        its = C.c_int32(0)
        self._chk(self._lib.sw_eig_solve(self._h, src, dst, mode, float(tol), int(maxiter), C.byref(its)),
                  "sw_eig_solve")
        return int(its.value)

    def eig_gram(self, a, b):
        out = np.empty((64, 64), dtype=np.complex128)
        self._chk(self._lib.sw_eig_gram(self._h, a, b, _ptr(out)), "sw_eig_gram")
        return out

    def eig_rotate(self, src, Y, dst, sub=-1):
        """buf_dst = buf_src Y, or (sub >= 0) buf_dst = buf_sub - buf_src Y"""
        Y = np.ascontiguousarray(Y, dtype=np.complex128)
        if Y.shape != (64, 64):
            raise EngineError("rotation matrix must be 64 x 64")
        self._chk(self._lib.sw_eig_rotate(self._h, src, _ptr(Y), dst, sub), "sw_eig_rotate")

    def eig_fetch(self, src, k):
        out = np.empty((k, self._eig_n), dtype=np.complex128)
        self._chk(self._lib.sw_eig_fetch(self._h, src, k, _ptr(out)), "sw_eig_fetch")
        return out

    def eig_end(self):
        self._chk(self._lib.sw_eig_end(self._h), "sw_eig_end")

    # -- building blocks -------------------------------------------------------------------
    def _io(self, X, n_in):
        X = _c128(X)
        single = X.ndim == 1
        X2 = X.reshape(1, -1) if single else X
        if X2.shape[1] != n_in:
            raise EngineError("vector length %d, expected %d" % (X2.shape[1], n_in))
        return X2, single

    def _n(self, hid, level):
        return self.level_sizes[hid][level]

    def apply_dirac(self, hid, level, X):
        X2, single = self._io(X, self._n(hid, level))
        Y = np.empty_like(X2)
        self._chk(self._lib.sw_apply_dirac(self._h, hid, level, X2.shape[0], _ptr(X2), _ptr(Y)),
                  "sw_apply_dirac")
        return Y[0] if single else Y

    def restrict(self, hid, level, X):
        X2, single = self._io(X, self._n(hid, level))
        Y = np.empty((X2.shape[0], self._n(hid, level + 1)), dtype=np.complex128)
        self._chk(self._lib.sw_restrict(self._h, hid, level, X2.shape[0], _ptr(X2), _ptr(Y)),
                  "sw_restrict")
        return Y[0] if single else Y

    def prolong(self, hid, level, X):
        X2, single = self._io(X, self._n(hid, level + 1))
        Y = np.empty((X2.shape[0], self._n(hid, level)), dtype=np.complex128)
        self._chk(self._lib.sw_prolong(self._h, hid, level, X2.shape[0], _ptr(X2), _ptr(Y)),
                  "sw_prolong")
        return Y[0] if single else Y

    def coarsest(self, hid, X):
        X2, single = self._io(X, self.level_sizes[hid][-1])
        Y = np.empty_like(X2)
        self._chk(self._lib.sw_coarsest(self._h, hid, X2.shape[0], _ptr(X2), _ptr(Y)),
                  "sw_coarsest")
        return Y[0] if single else Y

    def vcycle(self, hid, level0, B):
        B2, single = self._io(B, self._n(hid, level0))
        X = np.empty_like(B2)
        self._chk(self._lib.sw_vcycle(self._h, hid, level0, B2.shape[0], _ptr(B2), _ptr(X)),
                  "sw_vcycle")
        return X[0] if single else X

    def solve(self, hid, level0, B, tol, maxiter=1000):
        B2, single = self._io(B, self._n(hid, level0))
        nb = B2.shape[0]
        X = np.empty_like(B2)
        iters = np.zeros(nb, dtype=np.int32)
        relres = np.zeros(nb, dtype=np.float64)
        self._chk(self._lib.sw_solve(self._h, hid, level0, nb, _ptr(B2), _ptr(X), float(tol),
                                     int(maxiter), _ptr(iters), _ptr(relres)), "sw_solve")
        if single:
            return X[0], int(iters[0]), float(relres[0])
        return X, iters, relres

    # -- probe batches ---------------------------------------------------------------------
    @staticmethod
    def _probes(probes):
        p = np.ascontiguousarray(probes, dtype=np.int8)
        if p.ndim == 1:
            p = p.reshape(1, -1)
        return p

    def hutch_batch(self, mode, level, probes, tol, maxiter=1000):
        p = self._probes(probes)
        nb = p.shape[0]
        ests = np.zeros(nb, dtype=np.complex128)
        iters = np.zeros(2 * nb, dtype=np.int32)
        self._chk(self._lib.sw_hutch_batch(self._h, mode, level, nb, _ptr(p), float(tol),
                                           int(maxiter), _ptr(ests), _ptr(iters)),
                  "sw_hutch_batch")
        return ests, iters[:nb].copy(), iters[nb:].copy()

    def probes_upload(self, level, probes):
        p = self._probes(probes)
        self._nb_uploaded = p.shape[0]
        self._chk(self._lib.sw_probes_upload(self._h, level, p.shape[0], _ptr(p)),
                  "sw_probes_upload")

    def probes_upload_slot(self, slot, level, probes):
        p = self._probes(probes)
        self._chk(self._lib.sw_probes_upload_slot(self._h, slot, level, p.shape[0], _ptr(p)),
                  "sw_probes_upload_slot")
        self._slot_nb = getattr(self, "_slot_nb", {})
        self._slot_nb[slot] = p.shape[0]

    def probes_select(self, slot):
        self._chk(self._lib.sw_probes_select(self._h, slot), "sw_probes_select")
        self._nb_uploaded = self._slot_nb[slot]

    def stream_set(self, window):
        """Hand the engine a window of the MT19937 probe stream (ProbeStream.window()); it
        becomes stream position 0 of probes_generate()."""
        w = np.ascontiguousarray(window, dtype=np.uint32)
        if w.size != 624:
            raise EngineError("an MT19937 window has 624 words")
        self._chk(self._lib.sw_probes_stream_set(self._h, _ptr(w)), "sw_probes_stream_set")

    def probes_generate(self, slot, level, nb, pos, kind="z2"):
        """Fill `slot` on the device with the nb probes that start at draw `pos` of the stream."""
        self._chk(self._lib.sw_probes_generate(self._h, int(slot), int(level), int(nb),
                                               PROBE_KINDS[kind], int(pos)), "sw_probes_generate")
        self._slot_nb = getattr(self, "_slot_nb", {})
        self._slot_nb[slot] = int(nb)
        self._slot_n = getattr(self, "_slot_n", {})
        self._slot_n[slot] = self.level_sizes[0][level]

    def probes_fetch(self, slot):
        """int8 codes held in `slot`, shape (nb, n) (tests; the estimators never need them)."""
        nb = self._slot_nb[slot]
        n = getattr(self, "_slot_n", {}).get(slot)
        if n is None:
            raise EngineError("slot %d was not generated on the device" % slot)
        out = np.empty((nb, n), dtype=np.int8)
        self._chk(self._lib.sw_probes_fetch(self._h, int(slot), _ptr(out)), "sw_probes_fetch")
        return out

    def kernel_stats(self, which):
        ms = C.c_double(0.0)
        cnt = C.c_int64(0)
        self._chk(self._lib.sw_kernel_stats(self._h, which, C.byref(ms), C.byref(cnt)),
                  "sw_kernel_stats")
        return ms.value, int(cnt.value)

    def kernel_work(self, which):
        w = C.c_double(0.0)
        self._chk(self._lib.sw_kernel_work(self._h, which, C.byref(w)), "sw_kernel_work")
        return w.value

    def hutch_run(self, mode, level, tol, maxiter=1000):
        self._chk(self._lib.sw_hutch_run(self._h, mode, level, float(tol), int(maxiter)),
                  "sw_hutch_run")

    def sync(self):
        self._chk(self._lib.sw_sync(self._h), "sw_sync")

    def hutch_fetch(self):
        nb = self._nb_uploaded
        ests = np.zeros(nb, dtype=np.complex128)
        iters = np.zeros(2 * nb, dtype=np.int32)
        self._chk(self._lib.sw_hutch_fetch(self._h, _ptr(ests), _ptr(iters)), "sw_hutch_fetch")
        return ests, iters[:nb].copy(), iters[nb:].copy()

    # -- the collective behind the C ABI (RCCL) -----------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        if load_library().sw_comm_unique_id(buf) != 0:
            raise EngineError("sw_comm_unique_id failed (RCCL not loadable)")
        return buf.raw

    def comm_init(self, nranks, rank, uid):
        buf = C.create_string_buffer(bytes(uid), 128)
        self._chk(self._lib.sw_comm_init(self._h, int(nranks), int(rank), buf), "sw_comm_init")

    def allreduce_stats(self, stats):
        a = np.ascontiguousarray(stats, dtype=np.float64).copy()
        if a.size != 4:
            raise EngineError("four statistics expected")
        self._chk(self._lib.sw_allreduce_stats(self._h, _ptr(a)), "sw_allreduce_stats")
        return a

    def comm_destroy(self):
        self._chk(self._lib.sw_comm_destroy(self._h), "sw_comm_destroy")

    # -- measurement -----------------------------------------------------------------------
    def bench_dirac(self, hid, level, nb, reps):
        ms = C.c_double(0.0)
        self._chk(self._lib.sw_bench_dirac(self._h, hid, level, nb, reps, C.byref(ms)),
                  "sw_bench_dirac")
        return ms.value

    def set_profiling(self, on):
        self._chk(self._lib.sw_set_profiling(self._h, 1 if on else 0), "sw_set_profiling")

    def timers(self):
        t = (C.c_double * 8)()
        self._chk(self._lib.sw_timers(self._h, t), "sw_timers")
        return dict(zip(TIMER_NAMES, [float(v) for v in t]))

    def timers_reset(self):
        self._chk(self._lib.sw_timers_reset(self._h), "sw_timers_reset")

    def launch_count(self):
        n = C.c_int64(0)
        self._chk(self._lib.sw_launch_count(self._h, C.byref(n)), "sw_launch_count")
        return int(n.value)


class ProbeStream:
    """The reference's probe stream (utils.py:213-216) without NumPy's global state: MT19937
    seeded as ``np.random.seed(seed)``; host-only, needs no GPU.  ``jump`` moves any distance in
    O(1) state refills (GF(2) jump polynomial), ``skip`` walks there (the checker)."""

    def __init__(self, seed=None, _handle=None):
        self._lib = load_library()
        if _handle is not None:
            self._g = _handle
        else:
            self._g = self._lib.sw_mt_create(int(seed) & 0xFFFFFFFF)
        if not self._g:
            raise EngineError("sw_mt_create failed")

    @classmethod
    def from_numpy_state(cls, state=None):
        """Continue the legacy global NumPy stream (np.random.get_state() by default)."""
        lib = load_library()
        st = np.random.get_state() if state is None else state
        if st[0] != 'MT19937':
            raise EngineError("not an MT19937 state")
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        g = lib.sw_mt_from_state(_ptr(key), int(st[2]))
        return cls(_handle=g)

    def numpy_state(self):
        """State tuple for np.random.set_state (same stream from here on)."""
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int(0)
        self._lib.sw_mt_get_state(self._g, _ptr(key), C.byref(pos))
        return ('MT19937', key, int(pos.value), 0, 0.0)

    def copy(self):
        return ProbeStream.from_numpy_state(self.numpy_state())

    def skip(self, ndraws):
        self._lib.sw_mt_skip(self._g, int(ndraws))

    def jump(self, ndraws):
        if self._lib.sw_mt_jump(self._g, int(ndraws)) != 0:
            raise EngineError("sw_mt_jump failed")

    def window(self):
        """The 624 raw words at the current position (Engine.stream_set)."""
        out = np.empty(624, dtype=np.uint32)
        self._lib.sw_mt_window(self._g, _ptr(out))
        return out

    def raw(self, n):
        out = np.empty(int(n), dtype=np.uint32)
        self._lib.sw_mt_raw(self._g, int(n), _ptr(out))
        return out

    def rademacher(self, count, n):
        out = np.empty((int(count), int(n)), dtype=np.int8)
        self._lib.sw_mt_rademacher(self._g, int(count) * int(n), _ptr(out))
        return out

    def z4(self, count, n):
        out = np.empty((int(count), int(n)), dtype=np.int8)
        self._lib.sw_mt_z4(self._g, int(count) * int(n), _ptr(out))
        return out

    def probes(self, count, n, kind="z2"):
        return self.rademacher(count, n) if kind == "z2" else self.z4(count, n)

    def __del__(self):  # pragma: no cover
        try:
            if self._g:
                self._lib.sw_mt_destroy(self._g)
                self._g = None
        except Exception:
            pass
