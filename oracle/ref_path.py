"""NumPy/SciPy restatement of the reference's Tr(A^-1) path (CPU oracle).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line (under /root/reference) it follows.  The reference itself
cannot be imported whole in this image (``multigrid.py:4`` imports ``pyamg``,
which is absent, version unpinned by the reference), so:

* everything that does not depend on pyamg is pinned by (a) the reference's
  only known-answer value, the "exact trace" comment at ``gateway.py:100-104``,
  and (b) golden vectors produced by importing the reference's own
  ``matrix.py`` / ``utils.py`` (they import fine) -- see
  ``tests/golden/make_golden.py``;
* the flexible-GMRES outer iteration restates pyamg's *published* algorithm
  (right-preconditioned flexible GMRES, x0 = 0, stop when ||r|| < tol*||b||);
  iteration COUNTS through that boundary are "parity unpinned".  Converged
  results do not depend on it (checked against sparse LU to ~1e-12).

Layout conventions are the reference's: vectors are flat complex128 arrays in
the spin-major order idx(s,x,y) = s*L^2 + y*L + x.
"""
import os
import time
from math import sqrt

import numpy as np
import scipy.io as sio
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.linalg import LinearOperator, lgmres, eigs, eigsh


# ----------------------------------------------------------------------------
# a1: matrix loading                                           matrix.py:14-31
# ----------------------------------------------------------------------------
def load_matrix(path, mass, unflip_g3=None):
    """matrix.py:14-31: S from the MATLAB file, gamma3 un-flip for the 16^2 file
    only (matrix.py:24-27, keyed on the *basename* 'schwinger16.mat'), A = S + m*I.

    Modern SciPy returns COO from loadmat, on which the reference's row-slice
    assignment raises; the restatement does the same arithmetic on CSR."""
    S = sp.csr_matrix(sio.loadmat(path)["S"]).astype(np.complex128)
    if unflip_g3 is None:
        unflip_g3 = os.path.basename(path) == "schwinger16.mat"
    if unflip_g3:
        half = S.shape[0] // 2
        sign = np.ones(S.shape[0])
        sign[half:] = -1.0
        S = sp.diags(sign) @ S
    A = S + mass * sp.identity(S.shape[0], dtype=S.dtype)
    return sp.csr_matrix(A)


def lattice_index(s, x, y, L):
    """idx(s,x,y) = s*L^2 + y*L + x."""
    return s * L * L + y * L + x


def extract_links(S, L):
    """U1(n) = -S[idx(0,n), idx(0,n+x)],  U2(n) = -S[idx(0,n), idx(0,n+y)]
    (periodic).  Returns two (L*L,) arrays indexed by y*L+x."""
    S = sp.csr_matrix(S)
    V = L * L
    n = np.arange(V)
    x, y = n % L, n // L
    xp = y * L + (x + 1) % L
    yp = ((y + 1) % L) * L + x
    U1 = -np.asarray(S[n, xp]).ravel()
    U2 = -np.asarray(S[n, yp]).ravel()
    return U1, U2


def build_wilson(U1, U2, L):
    """The stencil of SURVEY F2 as an explicit CSR matrix (S, no mass term):

    (S psi)_a(n) = 4 psi_a(n) - sum_b [ (1-s1)_ab U1(n) psi_b(n+x) + (1+s1)_ab U1*(n-x) psi_b(n-x)
                                      + (1-s2)_ab U2(n) psi_b(n+y) + (1+s2)_ab U2*(n-y) psi_b(n-y) ]
    """
    V = L * L
    n = np.arange(V)
    x, y = n % L, n // L
    xp = y * L + (x + 1) % L
    xm = y * L + (x - 1) % L
    yp = ((y + 1) % L) * L + x
    ym = ((y - 1) % L) * L + x
    one_m_s1 = np.array([[1, -1], [-1, 1]], dtype=np.complex128)
    one_p_s1 = np.array([[1, 1], [1, 1]], dtype=np.complex128)
    one_m_s2 = np.array([[1, 1j], [-1j, 1]], dtype=np.complex128)
    one_p_s2 = np.array([[1, -1j], [1j, 1]], dtype=np.complex128)
    rows, cols, vals = [], [], []
    for a in range(2):
        rows.append(a * V + n); cols.append(a * V + n); vals.append(np.full(V, 4.0 + 0j))
        for b in range(2):
            for (G, U, nb) in ((one_m_s1, U1, xp), (one_p_s1, np.conj(U1[xm]), xm),
                               (one_m_s2, U2, yp), (one_p_s2, np.conj(U2[ym]), ym)):
                rows.append(a * V + n); cols.append(b * V + nb); vals.append(-G[a, b] * U)
    M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(2 * V, 2 * V))
    return sp.csr_matrix(M)


# ----------------------------------------------------------------------------
# a15: reporting / plumbing                                   utils.py:19-125
# ----------------------------------------------------------------------------
def flops_v_manual(bare_level, levels, level_id, smooth_iters):
    """utils.py:19-31 (recursive nnz model)."""
    f = (2 * smooth_iters + (2 if level_id == bare_level else 1)) * levels[level_id].A.nnz
    if level_id == len(levels) - 2:
        return f
    return f + flops_v_manual(bare_level, levels, level_id + 1, smooth_iters)


# ----------------------------------------------------------------------------
# a2/a3: hierarchy                                       multigrid.py:26-48,100-345
# ----------------------------------------------------------------------------
class OLevel:
    """multigrid.py:26-37 LevelML."""
    def __init__(self):
        self.R = 0; self.P = 0; self.A = 0; self.Q = 0
        self.Pperm = 0; self.perm_shift = 0; self.Bblock_perm = 0; self.g3 = 0


def pperm_matrix(n, shift):
    """multigrid.py:151-153 / 324-326: the operator saved as `Pperm`."""
    diagonals = [np.ones(n - shift), np.ones(shift)]
    return sp.csr_matrix(sp.diags(diagonals, [-shift, n - shift]).transpose())


def build_P_from_testvectors(eig_vecs, n, i, dof, aggrs):
    """multigrid.py:192-259: scatter the test vectors into the block-column P
    with the even/odd 'spin' split, then per-aggregate classical Gram-Schmidt
    (single pass).  The reference does this on a dense n x m array; here the
    same index arithmetic runs on per-aggregate blocks."""
    dofi = dof[i] if i == 0 else int(dof[i] / 2)
    dofip1 = int(dof[i + 1] / 2)
    aggr_size = aggrs[i] * dofi if i == 0 else aggrs[i] * dofi * 2
    aggr_size_half = int(aggr_size / 2)
    nr_aggrs = int(n / aggr_size)
    m = nr_aggrs * dofip1 * 2
    blocks = np.zeros((nr_aggrs, aggr_size, 2 * dofip1), dtype=np.complex128)
    hd = int(dofi / 2)
    for j in range(nr_aggrs):
        for k in range(dofip1):
            for w in range(int(aggr_size_half / (dofi / 2))):
                for z in range(hd):
                    # "spin 0" entries, multigrid.py:205-215
                    loc = w * dofi + z
                    blocks[j, loc, k] = eig_vecs[j * aggr_size + loc, k]
                    # "spin 1" entries, multigrid.py:218-227
                    loc = w * dofi + hd + z
                    blocks[j, loc, dofip1 + k] = eig_vecs[j * aggr_size + loc, k]
    # multigrid.py:232-259: plain CGS, column by column, both halves
    for j in range(nr_aggrs):
        for off in (0, dofip1):
            for k in range(dofip1):
                col = blocks[j, :, off + k]
                rs = [np.vdot(blocks[j, :, off + w], col) for w in range(k)]
                for w in range(k):
                    col -= rs[w] * blocks[j, :, off + w]
                col /= sqrt(np.vdot(col, col).real)
    rows = (np.arange(nr_aggrs)[:, None, None] * aggr_size + np.arange(aggr_size)[None, :, None]
            + np.zeros((1, 1, 2 * dofip1), dtype=np.int64))
    cols = (np.arange(nr_aggrs)[:, None, None] * (2 * dofip1) + np.arange(2 * dofip1)[None, None, :]
            + np.zeros((1, aggr_size, 1), dtype=np.int64))
    nzmask = blocks != 0
    P = sp.csr_matrix((blocks[nzmask], (rows[nzmask], cols[nzmask])), shape=(n, m))
    return P


def mg_setup(A, dof, aggrs, max_levels, acc_eigvs, params, testvectors=None):
    """multigrid.py:100-345.  `testvectors` (list per level) overrides the
    ARPACK call so a fixed hierarchy can be reproduced (which
    member of a complex-conjugate eigenpair ARPACK returns is not pinned)."""
    Al = sp.csr_matrix(A).copy()
    levels = [OLevel()]
    levels[0].A = Al.copy()
    used_tv = []
    for i in range(max_levels - 1):
        n = Al.shape[0]
        dofip1 = int(dof[i + 1] / 2)
        diag_g3 = np.ones(n); diag_g3[n // 2:] = -1.0           # :130-133
        levels[i].g3 = sp.diags([diag_g3], [0])
        if params["use_permuted"] and i == 0:                     # :142-155
            mat_disp = params["latt_dims"][0] * 2 * params["x_displacement"]
            levels[0].perm_shift = mat_disp
            levels[0].Pperm = pperm_matrix(n, mat_disp)
            levels[0].Bblock_perm = sp.identity(n, dtype=Al.dtype, format="csr")
        if acc_eigvs == "low":                                    # :164-171
            tolx, ncvx = 1.0e-3, dofip1 + 2
        elif acc_eigvs == "high":
            tolx, ncvx = 1.0e-9, None
        else:
            raise Exception("<accuracy_mg_eigvs> does not have a possible value.")
        if params["test_vectors_type"] != "EVs":
            raise Exception("oracle restates the 'EVs' test-vector mode only")
        if testvectors is not None:
            eig_vecs = testvectors[i]
        else:                                                     # :174
            _, eig_vecs = eigs(sp.csc_matrix(Al), k=dofip1, which="LM", tol=tolx,
                               maxiter=1000000, sigma=0.0, ncv=ncvx)
        used_tv.append(eig_vecs)
        Pl = build_P_from_testvectors(eig_vecs, n, i, dof, aggrs)  # :192-262
        Rl = sp.csr_matrix(Pl.conjugate().transpose())            # :267-274
        levels[i].P = Pl
        levels[i].R = Rl
        Al = sp.csr_matrix(Rl @ Al @ Pl)                         # :276
        levels.append(OLevel())
        levels[i + 1].A = Al.copy()
        if params["use_permuted"]:                                # :320-331
            mat_disp = int((levels[i].perm_shift / (dof[i] * aggrs[i])) * dof[i + 1])
            levels[i + 1].perm_shift = mat_disp
            levels[i + 1].Pperm = pperm_matrix(Pl.shape[1], mat_disp)
            Bl = levels[i].Pperm.transpose().conjugate() @ (Pl @ levels[i + 1].Pperm)
            Bl = (Rl @ levels[i].Bblock_perm) @ Bl
            levels[i + 1].Bblock_perm = sp.csr_matrix(Bl)
    coarsest_inv = np.linalg.inv(levels[-1].A.toarray())          # :342-344
    return levels, coarsest_inv, used_tv


# ----------------------------------------------------------------------------
# a9: flexible GMRES (pyamg.krylov.fgmres call site multigrid.py:362)
# ----------------------------------------------------------------------------
def fgmres(matvec, b, tol, precond, maxiter):
    """Right-preconditioned flexible GMRES, x0 = 0, no restart, stop when
    ||r|| < tol*||b||.  Restates the published pyamg.krylov.fgmres algorithm
    (pyamg is absent; version unpinned) with MGS Arnoldi + Givens in place of
    Householder.  Returns (x, iterations); b = 0 returns zeros."""
    n = b.shape[0]
    normb = np.linalg.norm(b)
    x = np.zeros(n, dtype=np.complex128)
    if normb == 0.0:
        return x, 0
    V = [b / normb]
    Z = []
    H = np.zeros((maxiter + 1, maxiter), dtype=np.complex128)
    cs = np.zeros(maxiter, dtype=np.complex128)
    sn = np.zeros(maxiter, dtype=np.complex128)
    g = np.zeros(maxiter + 1, dtype=np.complex128)
    g[0] = normb
    its = 0
    for j in range(maxiter):
        z = precond(V[j])
        w = matvec(z)
        Z.append(z)
        for k in range(j + 1):
            H[k, j] = np.vdot(V[k], w)
            w = w - H[k, j] * V[k]
        H[j + 1, j] = np.linalg.norm(w)
        if H[j + 1, j] != 0:
            V.append(w / H[j + 1, j])
        else:
            V.append(w)
        for k in range(j):
            t = cs[k] * H[k, j] + sn[k] * H[k + 1, j]
            H[k + 1, j] = -np.conj(sn[k]) * H[k, j] + cs[k] * H[k + 1, j]
            H[k, j] = t
        a, bb = H[j, j], H[j + 1, j]
        d = sqrt(abs(a) ** 2 + abs(bb) ** 2)
        cs[j], sn[j] = (abs(a) / d, (a / abs(a)) * np.conj(bb) / d) if a != 0 else (0.0, 1.0)
        H[j, j] = cs[j] * a + sn[j] * bb
        H[j + 1, j] = 0.0
        g[j + 1] = -np.conj(sn[j]) * g[j]
        g[j] = cs[j] * g[j]
        its = j + 1
        if abs(g[j + 1]) < tol * normb:
            break
    y = np.linalg.solve(np.triu(H[:its, :its]), g[:its])
    for k in range(its):
        x += y[k] * Z[k]
    return x, its


class Timer:
    """utils.py:366-445 CustomTimer (non-reentrant wall-clock buckets)."""
    PARTS = ("mvm", "defl", "P", "R", "mg_setup", "defl_setup", "axpy")

    def __init__(self):
        self.on = 0
        self.reset()

    def reset(self):
        for p in self.PARTS:
            setattr(self, p, 0.0)
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
        if part not in self.PARTS:
            raise Exception("Uknown part to time")
        setattr(self, part, getattr(self, part) + time.time() - self.tbuff)


class OracleMG:
    """multigrid.py:56-557 class MG, same attribute names (a2)."""

    def __init__(self, A, smooth_iters=2):
        self.level_nr = 0
        self.ml_levels = []
        self.A = A
        self.x = []
        self.num_iters = 0
        self.total_levels = 0
        self.smooth_iters = smooth_iters
        self.coarsest_lev_iters = [0] * 10
        self.level_for_diff_op = 0
        self.solve_tol = 1.0e-1
        self.coarsest_inv = []
        self.timer = Timer()
        self.skip_level = False
        self.nr_vcycles = 0
        self.nr_matvecs = 0

    class _ML:
        pass

    def setup(self, dof, aggrs, max_levels, acc_eigvs, params, testvectors=None):
        levels, cinv, tv = mg_setup(self.A, dof, aggrs, max_levels, acc_eigvs, params, testvectors)
        self.ml = OracleMG._ML()
        self.ml.levels = levels
        self.coarsest_inv = cinv
        self.testvectors = tv
        self.total_levels = len(levels)

    def matvec(self, x):                                          # multigrid.py:552-557
        self.nr_matvecs += 1
        return self.A @ x

    def solve(self, A, b, tol):                                   # multigrid.py:347-366
        maxiter = A.shape[0] if A.shape[0] < 1000 else 1000
        self.A = self.ml.levels[self.level_nr].A
        # LinearOperator(shape, matvec=...) probes matvec(zeros) once
        lop1 = LinearOperator(A.shape, matvec=self.matvec)
        lop2 = LinearOperator(A.shape, matvec=self.one_mg_step)
        self.x, self.num_iters = fgmres(lop1.matvec, b, tol, lop2.matvec, maxiter)

    def one_mg_step(self, b):                                     # multigrid.py:369-447
        self.nr_vcycles += 1
        lv = self.ml.levels
        l0 = self.level_nr
        level_id = self.total_levels - l0
        bs = [None] * level_id
        xs = [np.zeros(lv[i].A.shape[0], dtype=np.complex128) for i in range(l0, self.total_levels)]
        bs[0] = b.copy()
        i = -1
        for i in range(level_id - 1):
            Ai = lv[i + l0].A
            r = bs[i] - Ai @ xs[i]                                # :388
            self.A = Ai
            lop = LinearOperator(Ai.shape, matvec=self.matvec)    # :392
            e, _ = lgmres(lop, r, rtol=1.0e-20, atol=0.0, maxiter=self.smooth_iters)   # :393
            self.A = lv[l0].A
            xs[i] = xs[i] + e
            r = bs[i] - Ai @ xs[i]                                # :402
            bs[i + 1] = lv[i + l0].R @ r                          # :406
        i += 1
        xs[i] = np.asarray(np.dot(self.coarsest_inv, bs[i])).reshape(-1)   # :414-415
        self.coarsest_lev_iters[l0] += 1
        for i in range(level_id - 2, -1, -1):
            Ai = lv[i + l0].A
            xs[i] = xs[i] + lv[i + l0].P @ xs[i + 1]              # :429
            r = bs[i] - Ai @ xs[i]                                # :433
            self.A = Ai
            lop = LinearOperator(Ai.shape, matvec=self.matvec)
            e, _ = lgmres(lop, r, rtol=1.0e-20, atol=0.0, maxiter=self.smooth_iters)   # :438
            self.A = lv[l0].A
            xs[i] = xs[i] + e
        return xs[0]


# ----------------------------------------------------------------------------
# a12: deflation                                               utils.py:130-201
# ----------------------------------------------------------------------------
def deflation_hutchinson(A, g3, Pperm, k, tolx, use_permuted, eigvecs=None):
    """utils.py:135-140,145-155,170-173,191: eigenpairs of Q = g3*A nearest 0,
    U = Pperm*g3*V*sgn, tr1 = sum_i (u_i^H v_i)/|lambda_i|.
    Returns (Ux, tr1, Vx, Sy)."""
    if k <= 0:
        return None, 0.0, None, None
    Q = sp.csc_matrix(g3 @ A)
    if eigvecs is None:
        Sy, Vx = eigsh(Q, k=k, which="LM", tol=tolx, sigma=0.0)
    else:
        Sy, Vx = eigvecs
    sgn = np.where(Sy > 0, 1.0, -1.0)
    Sabs = Sy * sgn
    Ux = Vx * sgn[None, :]
    Ux = g3 @ Ux
    if use_permuted:
        Ux = Pperm @ Ux
    small = np.dot(Ux.conj().T, Vx) * np.linalg.inv(np.diag(Sabs))   # elementwise, :173
    return Ux, np.trace(small), Vx, Sy


def rademacher(n):
    """utils.py:213-216 / 255-258: global NumPy stream, one draw per entry."""
    x = np.random.randint(2, size=n)
    x *= 2
    x -= 1
    return x.astype(np.complex128)


# ----------------------------------------------------------------------------
# a10/a11: one probe                                           utils.py:207-361
# ----------------------------------------------------------------------------
def hutch_probe(x, solve, Ux, PpermT):
    """utils.py:221-249 given the probe x and a solve(b)->A^-1 b callable."""
    x_def = x - Ux @ (Ux.conj().T @ x) if Ux is not None else x
    rhs = PpermT @ x_def if PpermT is not None else x_def
    z = solve(rhs)
    return np.vdot(x, z)


def mlmc_probe(x0, i, levels, skip_level, solve_level, coarsest_inv, use_permuted, Vx=None):
    """utils.py:260-355; Vx = MLMC-level deflation vectors (utils.py:266) or None.
    solve_level(l, b) -> A_l^-1 b."""
    nlev = len(levels)
    x_def = x0 if Vx is None else x0 - Vx @ (Vx.conj().T @ x0)
    if use_permuted:
        x_perm = levels[i].Pperm.transpose() @ x_def
        x_def = levels[i].Bblock_perm @ x_perm
    z = solve_level(i, x_def)
    if skip_level and i == 0:
        xc = levels[1].R @ (levels[0].R @ x_def)
        lc = i + 2
    else:
        xc = levels[i].R @ x_def
        lc = i + 1
    if lc == nlev - 1:
        y = np.asarray(np.dot(coarsest_inv, xc)).reshape(-1)
    else:
        y = solve_level(lc, xc)
    if skip_level and i == 0:
        w = levels[0].P @ (levels[1].P @ y)
    else:
        w = levels[i].P @ y
    return np.vdot(x0, z) - np.vdot(x0, w)


def stopping_rule(ests, level_tol, min_index=5):
    """stoch_trace.py:137-154 / 386-406: sequential replay over per-probe values.
    Returns (index_i_at_break, ests_avg, ests_dev).  nr_ests = i (an index)."""
    n = len(ests)
    for i in range(n):
        cur = ests[: i + 1]
        avg = np.sum(cur) / (i + 1)
        dev = sqrt(np.sum(np.square(np.abs(cur - avg))) / (i + 1))
        if i >= min_index and dev / sqrt(i + 1) < level_tol:
            return i, avg, dev
    return n - 1, avg, dev


# ----------------------------------------------------------------------------
# exact (direct) helpers used to pin the per-probe values (SURVEY 8c tier 1)
# ----------------------------------------------------------------------------
class LUSolver:
    def __init__(self, A):
        self.lu = spla.splu(sp.csc_matrix(A))

    def __call__(self, b):
        return self.lu.solve(np.asarray(b, dtype=np.complex128))


def exact_trace_inverse(A, PpermT=None, block=512):
    """Tr(A^-1 * Pperm^T) by sparse LU, column blocks."""
    lu = spla.splu(sp.csc_matrix(A))
    n = A.shape[0]
    tr = 0.0 + 0.0j
    M = sp.identity(n, dtype=np.complex128, format="csc") if PpermT is None else sp.csc_matrix(PpermT)
    for c0 in range(0, n, block):
        c1 = min(n, c0 + block)
        X = lu.solve(M[:, c0:c1].toarray())
        tr += np.trace(X[c0:c1, :])
    return tr
