"""Host-side construction of the operands the HIP engine consumes.

Two hierarchies are built on the host (setup time, SciPy ARPACK + SuperLU as in the
reference) and uploaded once:

* the REFERENCE hierarchy -- index arithmetic of ``multigrid.py:100-345`` reproduced
  exactly (strip aggregates of consecutive rows, even/odd "spin" split, single-pass
  per-aggregate Gram-Schmidt, Galerkin coarse operators, Pperm / Bblock_perm
  bookkeeping).  It defines the MLMC level operators, so it must match the reference;
* an optional SOLVER hierarchy for level 0 -- 2-D site aggregates with a chirality
  split, used only to precondition level-0 solves.  Converged solves do not depend on
  the preconditioner, so this is free to differ from the reference.
"""
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def dense_inverse(M):
    """np.linalg.inv (multigrid.py:342-344).  The reference runs under OMP_NUM_THREADS=1
    (main.py:20); a coarsest level of several thousand unknowns (BASELINE config 2: 8192) would take
    minutes on one core, so large inversions borrow all host cores for the duration of the call."""
    M = np.asarray(M)
    if M.shape[0] >= 2048:
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=os.cpu_count() or 1):
                return np.linalg.inv(M)
        except ImportError:
            pass
    return np.linalg.inv(M)


# ---------------------------------------------------------------------------------------
# lattice helpers
# ---------------------------------------------------------------------------------------
def links_from_matrix(S, L):
    """U1(n) = -S[idx(0,n), idx(0,n+x)], U2(n) = -S[idx(0,n), idx(0,n+y)]."""
    S = sp.csr_matrix(S)
    site = np.arange(L * L)
    x, y = site % L, site // L
    right = y * L + (x + 1) % L
    up = ((y + 1) % L) * L + x
    U1 = -np.asarray(S[site, right]).ravel()
    U2 = -np.asarray(S[site, up]).ravel()
    return U1.astype(np.complex128), U2.astype(np.complex128)


_HOP = {  # spin matrices of the four hops: (dx, dy, forward?) -> 2x2
    "+x": np.array([[1, -1], [-1, 1]], dtype=np.complex128),
    "-x": np.array([[1, 1], [1, 1]], dtype=np.complex128),
    "+y": np.array([[1, 1j], [-1j, 1]], dtype=np.complex128),
    "-y": np.array([[1, -1j], [1j, 1]], dtype=np.complex128),
}


def wilson_from_links(U1, U2, L):
    """Explicit CSR of S (diagonal 4, no mass) from U(1) links, ordering idx(s,x,y)=s*L^2+y*L+x."""
    V = L * L
    site = np.arange(V)
    x, y = site % L, site // L
    nbr = {
        "+x": (y * L + (x + 1) % L, U1),
        "-x": (y * L + (x - 1) % L, None),
        "+y": (((y + 1) % L) * L + x, U2),
        "-y": (((y - 1) % L) * L + x, None),
    }
    nbr["-x"] = (nbr["-x"][0], np.conj(U1[nbr["-x"][0]]))
    nbr["-y"] = (nbr["-y"][0], np.conj(U2[nbr["-y"][0]]))
    ri, ci, vv = [], [], []
    for a in range(2):
        ri.append(a * V + site)
        ci.append(a * V + site)
        vv.append(np.full(V, 4.0, dtype=np.complex128))
        for b in range(2):
            for key, (dst, link) in nbr.items():
                ri.append(a * V + site)
                ci.append(b * V + dst)
                vv.append(-_HOP[key][a, b] * link)
    M = sp.coo_matrix((np.concatenate(vv), (np.concatenate(ri), np.concatenate(ci))),
                      shape=(2 * V, 2 * V))
    return sp.csr_matrix(M)


def detect_lattice(A):
    """If A = S + m*I for a U(1) Wilson stencil on an L x L torus, return (L, mass, U1, U2);
    otherwise None.  The check is exact (entry-wise) so a positive answer means the matrix-free
    stencil reproduces A bit for bit."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    if n % 2:
        return None
    L = int(round(np.sqrt(n // 2)))
    if 2 * L * L != n or L < 4 or L % 2:
        return None
    d = A.diagonal()
    if not np.all(d == d[0]) or d[0].imag != 0.0:
        return None
    mass = float(d[0].real) - 4.0
    S = A - sp.identity(n, dtype=A.dtype, format="csr") * (mass)
    U1, U2 = links_from_matrix(S, L)
    if abs(np.abs(U1) - 1).max() > 1e-12 or abs(np.abs(U2) - 1).max() > 1e-12:
        return None
    S2 = wilson_from_links(U1, U2, L)
    A2 = S2 + sp.identity(n, dtype=np.complex128, format="csr") * mass
    diff = abs(A - A2)
    if diff.nnz and diff.max() > 0.0:
        return None
    return L, mass, U1, U2


# ---------------------------------------------------------------------------------------
# reference hierarchy
# ---------------------------------------------------------------------------------------
class LevelML:
    """Per-level container with the attribute names ``stoch_trace``/``utils`` read
    (multigrid.py:26-37)."""
    R = 0
    P = 0
    A = 0
    Q = 0
    Pperm = 0
    perm_shift = 0
    Bblock_perm = 0
    g3 = 0


class SimpleML:
    def __init__(self):
        self.levels = []

    def __str__(self):
        out = []
        for idx, level in enumerate(self.levels[:-1]):
            out.append("Level: %d" % idx)
            out.append("\tsize(R) = " + str(level.R.shape))
            out.append("\tsize(P) = " + str(level.P.shape))
            out.append("\tsize(A) = " + str(level.A.shape))
        return "\n".join(out)


def shift_operator(n, shift):
    """The matrix the reference stores as ``Pperm`` (multigrid.py:151-153): Pperm^T moves
    entry i to i+shift (cyclic), i.e. (Pperm^T v)[i] = v[(i-shift) mod n]."""
    cols = np.arange(n)
    rows = (cols - shift) % n
    # Pperm[r, c] = 1 with r = (c - shift) mod n  <=>  Pperm^T[c, r] = 1
    return sp.csr_matrix((np.ones(n), (rows, cols)), shape=(n, n))


def level_layout(i, dof, aggrs):
    """Sizes used at level i of the reference setup (multigrid.py:123-127,192-199)."""
    dofi = dof[i] if i == 0 else int(dof[i] / 2)
    nvec = int(dof[i + 1] / 2)
    aggr_size = aggrs[i] * dofi if i == 0 else aggrs[i] * dofi * 2
    return dofi, nvec, aggr_size


def prolongator_from_testvectors(tv, n, i, dof, aggrs):
    """Block-column P of multigrid.py:192-262, vectorised over aggregates.

    Row ``j*aggr_size + loc`` belongs to aggregate j; within the aggregate the position
    ``loc % dofi`` decides the half: the first ``dofi/2`` positions of every group of
    ``dofi`` rows feed columns ``[0, nvec)`` of the aggregate, the rest ``[nvec, 2 nvec)``.
    Each half is orthonormalised with ONE classical Gram-Schmidt sweep (coefficients taken
    against the not-yet-updated column), as the reference does."""
    dofi, nvec, aggr_size = level_layout(i, dof, aggrs)
    if n % aggr_size:
        raise Exception("matrix size %d is not a multiple of the aggregate size %d" % (n, aggr_size))
    na = n // aggr_size
    hd = max(1, dofi // 2)
    loc = np.arange(aggr_size)
    upper = (loc % dofi) >= hd                       # True -> second ("spin 1") half
    blocks = np.asarray(tv)[:, :nvec].reshape(na, aggr_size, nvec).astype(np.complex128)
    halves = []
    for mask in (~upper, upper):
        Bh = blocks * mask[None, :, None]
        for k in range(nvec):
            if k:
                coef = np.einsum("arw,ar->aw", Bh[:, :, :k].conj(), Bh[:, :, k])
                Bh[:, :, k] -= np.einsum("arw,aw->ar", Bh[:, :, :k], coef)
            nrm = np.sqrt(np.einsum("ar,ar->a", Bh[:, :, k].conj(), Bh[:, :, k]).real)
            Bh[:, :, k] /= nrm[:, None]
        halves.append(Bh)
    full = np.concatenate(halves, axis=2)            # [na, aggr_size, 2 nvec]
    rows = (np.arange(na)[:, None, None] * aggr_size + loc[None, :, None]
            + np.zeros((1, 1, 2 * nvec), dtype=np.int64))
    cols = (np.arange(na)[:, None, None] * (2 * nvec) + np.arange(2 * nvec)[None, None, :]
            + np.zeros((1, aggr_size, 1), dtype=np.int64))
    keep = np.concatenate([np.broadcast_to((~upper)[None, :, None], (na, aggr_size, nvec)),
                           np.broadcast_to(upper[None, :, None], (na, aggr_size, nvec))], axis=2)
    P = sp.csr_matrix((full[keep], (rows[keep], cols[keep])), shape=(n, na * 2 * nvec))
    return P


def reference_hierarchy(A, dof, aggrs, max_levels, acc_eigvs, params, testvectors=None, eigs_fn=None,
                        invert=True):
    """multigrid.py:100-345 -> (SimpleML, coarsest_inv, testvectors).

    eigs_fn (optional): callable(level, A_level, nvec, tol) -> test vectors [n, nvec] or None -- the device
    eigensolver's hook for `eigs(A_l, k, sigma=0)` (multigrid.py:174); None (or a None answer) = the
    reference's own ARPACK + SuperLU call on the host.  invert = False leaves the dense inverse of the
    coarsest operator (multigrid.py:342-344) to the engine (sw_setup_invert_coarsest) and returns None for it."""
    tv_type = params["test_vectors_type"]
    if tv_type not in ("EVs", "LSVs", "RSVs"):
        raise Exception("unknown type of test vectors")
    if acc_eigvs == "low":
        tolx = 1.0e-3
    elif acc_eigvs == "high":
        tolx = 1.0e-9
    else:
        raise Exception("<accuracy_mg_eigvs> does not have a possible value.")
    ml = SimpleML()
    Al = sp.csr_matrix(A).astype(np.complex128)
    lev = LevelML()
    lev.A = Al.copy()
    ml.levels.append(lev)
    used = []
    for i in range(max_levels - 1):
        n = Al.shape[0]
        nvec = int(dof[i + 1] / 2)
        sign = np.ones(n)
        sign[n // 2:] = -1.0
        ml.levels[i].g3 = sp.diags([sign], [0])
        if params["use_permuted"] and i == 0:
            shift0 = params["latt_dims"][0] * 2 * params["x_displacement"]
            ml.levels[0].perm_shift = shift0
            ml.levels[0].Pperm = shift_operator(n, shift0)
            ml.levels[0].Bblock_perm = sp.identity(n, dtype=np.complex128, format="csr")
        tv = None
        if testvectors is not None and i < len(testvectors) and testvectors[i] is not None:
            tv = np.asarray(testvectors[i])
        elif tv_type == "EVs" and eigs_fn is not None:
            tv = eigs_fn(i, Al, nvec, tolx)
        if tv is not None:
            pass
        elif tv_type == "EVs":
            ncv = nvec + 2 if acc_eigvs == "low" else None
            _, tv = spla.eigs(sp.csc_matrix(Al), k=nvec, which="LM", tol=tolx, maxiter=1000000,
                              sigma=0.0, ncv=ncv)
        else:
            # singular vectors through the hermitian Q = g3 A (multigrid.py:159-188): its
            # eigenvectors are right singular vectors of A, g3 times them left singular vectors
            Q = sp.csc_matrix(ml.levels[i].g3 @ Al)
            _, tv = spla.eigsh(Q, k=nvec, which="LM", tol=tolx, sigma=0.0)
            tv = tv.astype(np.complex128)
            if tv_type == "LSVs":
                tv[n // 2:] = -tv[n // 2:]
        used.append(tv)
        Pl = prolongator_from_testvectors(tv, n, i, dof, aggrs)
        Rl = sp.csr_matrix(Pl.conjugate().transpose())
        ml.levels[i].P = Pl
        ml.levels[i].R = Rl
        Al = sp.csr_matrix(Rl @ Al @ Pl)
        nxt = LevelML()
        nxt.A = Al.copy()
        ml.levels.append(nxt)
        if params["use_permuted"]:
            sh = int((ml.levels[i].perm_shift / (dof[i] * aggrs[i])) * dof[i + 1])
            nxt.perm_shift = sh
            nxt.Pperm = shift_operator(Pl.shape[1], sh)
            Bl = ml.levels[i].Pperm.transpose().conjugate() @ (Pl @ nxt.Pperm)
            nxt.Bblock_perm = sp.csr_matrix((Rl @ ml.levels[i].Bblock_perm) @ Bl)
    coarsest_inv = np.matrix(dense_inverse(ml.levels[-1].A.toarray())) if invert else None
    return ml, coarsest_inv, used


# ---------------------------------------------------------------------------------------
# solver hierarchy (level-0 preconditioner only)
# ---------------------------------------------------------------------------------------
def weights_from_hessenberg(H):
    """Weights 1/theta_k of the fixed polynomial smoother from the (degree+1) x degree Hessenberg
    matrix of an Arnoldi run: theta_k are the roots of the GMRES(degree) residual polynomial (harmonic
    Ritz values), Leja-ordered for stability.  x <- x + w_k (b - A x), k = 0..degree-1."""
    import scipy.linalg as sla
    H = np.asarray(H, dtype=np.complex128)
    degree = H.shape[1]
    theta = list(sla.eig(H.conj().T @ H, H[:degree, :].conj().T)[0])
    ordered = [max(theta, key=abs)]
    theta.remove(ordered[0])
    while theta:
        nxt = max(theta, key=lambda t: np.prod([abs(t - o) for o in ordered]))
        ordered.append(nxt)
        theta.remove(nxt)
    return 1.0 / np.array(ordered, dtype=np.complex128)


def smoother_weights(A, degree, seed=2024, project=None):
    """Weights of a degree-`degree` fixed polynomial smoother for A (weights_from_hessenberg of an
    Arnoldi run on a random complex right-hand side; the device-built hierarchies run the Arnoldi
    process on the GPU instead: sw_setup_arnoldi).
    `project` (optional) maps the random start vector to the part of it the smoother is for (e.g.
    v - P R v: what the coarse correction leaves), so the polynomial spends its degree there."""
    if degree <= 0:
        return np.zeros(0, dtype=np.complex128)
    n = A.shape[0]
    rng = np.random.default_rng(seed)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    if project is not None:
        b = np.asarray(project(b), dtype=np.complex128)
    V = [b / np.linalg.norm(b)]
    H = np.zeros((degree + 1, degree), dtype=np.complex128)
    for j in range(degree):
        w = A @ V[j]
        for i in range(j + 1):
            H[i, j] = np.vdot(V[i], w)
            w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
    return weights_from_hessenberg(H)


def schur_complement(A, L):
    """Even-odd split of a level-0 operator in the reference order idx(s,x,y) = s L^2 + y L + x:
    returns (S, E, O, D) with S = D - A_eo A_oe / D on the even sites E (both spins), O the odd
    ones and D the constant diagonal (A_ee = A_oo = D I for the Wilson stencil)."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    V = L * L
    idx = np.arange(n)
    site = idx % V
    even = (((site % L) + (site // L)) & 1) == 0
    E, O = idx[even], idx[~even]
    D = A.diagonal()[0]
    Aeo = A[E][:, O]
    Aoe = A[O][:, E]
    S = sp.identity(E.size, dtype=np.complex128, format="csr") * D - (Aeo @ Aoe) / D
    return sp.csr_matrix(S), E, O, D


def matrix_from_block_rows(kcol, vals, n):
    """The MFMA block-row form of a level operator (kcol[RT, KS], vals[RT, KS, 64] with lane =
    (row & 15) + 16 (col & 3)) as a SciPy CSR matrix."""
    RT, KS = kcol.shape
    i = np.arange(16)
    c4 = np.arange(4)
    rows = (np.arange(RT)[:, None, None, None] * 16 + i[None, None, None, :]
            + np.zeros((1, KS, 4, 1), dtype=np.int64))
    cols = kcol[:, :, None, None].astype(np.int64) + c4[None, None, :, None] + np.zeros((1, 1, 1, 16), np.int64)
    data = np.asarray(vals).reshape(RT, KS, 4, 16)
    keep = data != 0
    return sp.csr_matrix((data[keep], (rows[keep], cols[keep])), shape=(n, n))


def block_rows_from_matrix(M, row_sites, n):
    """Pack the block rows `row_sites` (16 rows each) of an n x n sparse matrix whose non-zeros sit in
    16 x 16 site blocks into the MFMA block-row form: (tmap[RT], kcol[RT, KS], vals[RT, KS, 64])."""
    Mb = sp.bsr_matrix(sp.csr_matrix(M), blocksize=(16, 16))
    Mb.sort_indices()
    row_sites = np.asarray(row_sites, dtype=np.int64)
    nblk = np.diff(Mb.indptr)[row_sites]
    KB = max(1, int(nblk.max()))
    RT = row_sites.size
    kcol = np.zeros((RT, KB * 4), dtype=np.int32)
    vals = np.zeros((RT, KB * 4, 64), dtype=np.complex128)
    g = np.arange(4)
    for r, s_ in enumerate(row_sites):
        lo, hi = Mb.indptr[s_], Mb.indptr[s_ + 1]
        cs = Mb.indices[lo:hi]
        blk = Mb.data[lo:hi]                                   # [nb, 16 (i), 16 (col)]
        kcol[r, :] = int(s_) * 16                              # padding: a valid column, zero values
        nbk = hi - lo
        # the block of the row's own site goes LAST (after any padding): the smoother kernel then
        # finds its own X rows in the operand registers of the final four k-steps (k_bsr_mfma, xreg)
        own = np.flatnonzero(cs == s_)
        order = np.concatenate([np.flatnonzero(cs != s_), own])
        cs, blk = cs[order], blk[order]
        first = np.arange(nbk * 4)
        if own.size:
            first[(nbk - 1) * 4:] += (KB - nbk) * 4             # own block into the last four slots
        kcol[r, first] = (cs[:, None] * 16 + 4 * g[None, :]).reshape(-1)
        # vals[q*4+g, c4*16 + i] = blk[q, i, 4 g + c4]
        b4 = blk.reshape(nbk, 16, 4, 4)                         # [q, i, g, c4]
        vals[r, first, :] = b4.transpose(0, 2, 3, 1).reshape(nbk * 4, 64)
    return row_sites.astype(np.int32), kcol, vals


def coarse_schur_operators(A, Lc, even_site=None):
    """Even-odd Schur construction for a block level (5-point stencil of 16 x 16 blocks over an
    Lc x Lc site lattice, site-major rows; or any operator in 16-row tiles with a two-colouring
    `even_site` of the tiles under which equal colours do not couple).  Returns dict(S, F, G, Hb as sparse
    n x n matrices that are non-zero on the even / odd site rows only, E_sites, O_sites, E_rows, O_rows):
      S = D_ee - A_eo D_oo^-1 A_oe,  F = A_eo D_oo^-1,  G = D_oo^-1,  Hb = D_oo^-1 A_oe."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    if even_site is None:
        ns = Lc * Lc
        if n != ns * 16:
            raise Exception("block level of %d rows is not %d sites x 16" % (n, ns))
        site = np.arange(ns)
        even_site = (((site % Lc) + (site // Lc)) & 1) == 0
    else:
        even_site = np.asarray(even_site, dtype=bool)
        ns = even_site.size
        if n != ns * 16:
            raise Exception("operator of %d rows is not %d tiles x 16" % (n, ns))
        site = np.arange(ns)
    E_sites, O_sites = site[even_site], site[~even_site]
    row_even = np.repeat(even_site, 16)
    ME = sp.diags(row_even.astype(float))
    MO = sp.diags((~row_even).astype(float))
    Ab = sp.bsr_matrix(A, blocksize=(16, 16))
    Ab.sort_indices()
    # diagonal blocks of the odd sites, inverted
    dinv_blocks = np.zeros((O_sites.size, 16, 16), dtype=np.complex128)
    for q, s_ in enumerate(O_sites):
        lo, hi = Ab.indptr[s_], Ab.indptr[s_ + 1]
        k = lo + int(np.searchsorted(Ab.indices[lo:hi], s_))
        if k >= hi or Ab.indices[k] != s_:
            raise Exception("site %d has no diagonal block" % s_)
        dinv_blocks[q] = np.linalg.inv(Ab.data[k])
    G = sp.bsr_matrix((dinv_blocks, O_sites, np.concatenate([[0], np.cumsum((~even_site).astype(int))])),
                      shape=(n, n), blocksize=(16, 16)).tocsr()
    Aeo = (ME @ A @ MO).tocsr()
    Aoe = (MO @ A @ ME).tocsr()
    Aee = (ME @ A @ ME).tocsr()
    F = (Aeo @ G).tocsr()
    Hb = (G @ Aoe).tocsr()
    S = (Aee - Aeo @ Hb).tocsr()
    E_rows = np.nonzero(row_even)[0]
    O_rows = np.nonzero(~row_even)[0]
    return {"S": S, "F": F, "G": G, "Hb": Hb, "E_sites": E_sites, "O_sites": O_sites,
            "E_rows": E_rows, "O_rows": O_rows}


def reference_coarse_eo(P0, Ac, L, aggr_size):
    """Even-odd form of the COARSE level of a 2-level reference hierarchy on an L x L lattice (BASELINE
    config 2 as written: 32768 -> 8192), so that its exact solve costs a dense (n_c/2)^2 product instead of
    n_c^2.  The reference's aggregates (multigrid.py:192-262) are runs of `aggr_size` consecutive natural
    rows idx = s V + y L + x: strips of aggr_size sites along x at fixed (s, y), numbered
    j = s V/a + y L/a + xb, each with 2 nvec coarse dofs.  The two spin strips of one (y, xb) couple to each
    other and to the strips at (y, xb +- 1), (y +- 1, xb): with the coarse dofs reordered tile by tile,
    tile t = y L/a + xb = both strips of (y, xb) (16 rows when 2 nvec = 8), the coarse operator is a
    nearest-neighbour stencil of 16 x 16 blocks over an (L/a) x L torus and the colouring (y + xb) & 1 splits
    it (L/a even).  Returns (pi, packed) -- pi[new coarse index] = old, packed = the four operators S, F, G,
    Hb of the permuted A_c in MFMA block-row form for sw_set_eo_operator -- or None when the layout does not
    fit (other aggregate sizes, odd strip counts, couplings between equal colours)."""
    P0 = sp.csr_matrix(P0)
    Ac = sp.csr_matrix(Ac)
    n0, nc = P0.shape
    V = L * L
    if n0 != 2 * V or aggr_size <= 0 or L % aggr_size or n0 % aggr_size:
        return None
    na = n0 // aggr_size
    strips = L // aggr_size
    if nc % na or 2 * (nc // na) != 16 or strips % 2 or L % 2:
        return None
    per = nc // na
    nt = na // 2                                     # tiles = (y, xb) pairs
    t = np.arange(nt)
    new = ((t[:, None, None] * 2 + np.arange(2)[None, :, None]) * per + np.arange(per)[None, None, :])
    oldi = ((np.arange(2)[None, :, None] * nt + t[:, None, None]) * per + np.arange(per)[None, None, :])
    pi = np.empty(nc, dtype=np.int64)
    pi[new.reshape(-1)] = oldi.reshape(-1)
    even = (((t // strips) + (t % strips)) & 1) == 0
    Ap = Ac[pi][:, pi].tocsr()
    # equal colours must not couple (other than a tile with itself)
    Ab = sp.bsr_matrix(Ap, blocksize=(16, 16))
    rows = np.repeat(np.arange(nt), np.diff(Ab.indptr))
    cols = Ab.indices
    nz = np.abs(Ab.data).reshape(len(cols), -1).max(axis=1) > 0
    if np.any((even[rows] == even[cols]) & (rows != cols) & nz):
        return None
    ops = coarse_schur_operators(Ap, None, even_site=even)
    packed = [block_rows_from_matrix(ops["S"], ops["E_sites"], nc),
              block_rows_from_matrix(ops["F"], ops["E_sites"], nc),
              block_rows_from_matrix(ops["G"], ops["O_sites"], nc),
              block_rows_from_matrix(ops["Hb"], ops["O_sites"], nc)]
    return pi, packed


def auto_solver_cfg(L):
    """The solver configuration used when the caller names none: the tuned, device-built hierarchy where
    the lattice allows it (8 x 8 aggregates, then 2 x 2, the last smoothed level at least 8 x 8 sites:
    L a multiple of 16 and >= 128), the general default otherwise."""
    if L % 16 == 0 and L // 16 >= 8:
        return dict(TUNED_SOLVER_CFG_128)
    return dict(DEFAULT_SOLVER_CFG)


def eo_levels_of(cfg):
    """Levels smoothed on their even-odd Schur complement: cfg["eo_levels"], or [0] for the older
    switch cfg["eo_smoother"] = True."""
    lv = list(cfg.get("eo_levels", []))
    if cfg.get("eo_smoother") and 0 not in lv:
        lv.append(0)
    return sorted(lv)


def f32_capable(cfg):
    """Can the engine run this solver configuration's cycle in complex64 (option precond_f32)?  The
    complex64 path covers the fixed-polynomial cycles without pre-smoothing (K-cycles keep their small
    inner FGMRES in fp64), with the lattice level smoothed even-odd (the engine's own check is
    sw_engine.hip: f32_capable)."""
    if cfg.get("smoother", "richardson") != "richardson" or 0 not in eo_levels_of(cfg):
        return False
    return all(int(c[0]) == 0 for c in cfg["cycle"])


def site_blocks_of(A, Lc):
    """(nbr[ns, 5], blk[ns, 5, 16, 16]) of a block level's 5-point operator, or None when the matrix
    does not have exactly five 16 x 16 blocks in every block row (tiny lattices: general path)."""
    Ab = sp.bsr_matrix(sp.csr_matrix(A), blocksize=(16, 16))
    Ab.sort_indices()
    ns = Lc * Lc
    if Ab.shape[0] != ns * 16 or not np.all(np.diff(Ab.indptr) == 5):
        return None
    return Ab.indices.reshape(ns, 5).astype(np.int64), Ab.data.reshape(ns, 5, 16, 16)


def site_blocks_from_block_rows(kcol, vals):
    """The same from the engine's block-row form of a full level operator (sw_get_level_bsr: 20 k-steps
    per row tile = 5 site blocks x 4 column groups; vals[rt, 4 q + g, 16 c4 + i] = blk[rt, q, i, 4 g + c4])."""
    RT, KS = kcol.shape
    if KS != 20:
        return None
    nbr = (kcol[:, ::4] // 16).astype(np.int64)
    blk = np.asarray(vals).reshape(RT, 5, 4, 4, 16).transpose(0, 1, 4, 2, 3).reshape(RT, 5, 16, 16)
    return nbr, blk


def pack_site_blocks(blocks, targets):
    """blocks[R, nb, 16, 16] acting on the sites targets[R, nb] -> (kcol[R, 4 nb], vals[R, 4 nb, 64])
    of the MFMA block-row form (inverse of site_blocks_from_block_rows)."""
    R, nb = targets.shape
    g = np.arange(4)
    kcol = (targets[:, :, None] * 16 + 4 * g[None, None, :]).reshape(R, nb * 4).astype(np.int32)
    vals = blocks.reshape(R, nb, 16, 4, 4).transpose(0, 1, 3, 4, 2).reshape(R, nb * 4, 64)
    return kcol, np.ascontiguousarray(vals)


def coarse_schur_blocks(nbr, blk, Lc):
    """Even-odd Schur construction of a block level with batched 16 x 16 algebra (no sparse products,
    no per-site Python loops): the same S, F, G, Hb as coarse_schur_operators, directly in block-row
    form, for Lc >= 8 (displacements of +-2 must not alias on the periodic lattice).  Returns a dict with
    the packed operators [(tmap, kcol, vals) x 4], the site / row lists and S_ee as a sparse matrix
    over the even rows (for the smoother polynomial)."""
    ns = Lc * Lc
    site = np.arange(ns)
    xs, ys = site % Lc, site // Lc
    even = ((xs + ys) & 1) == 0
    is_self = nbr == site[:, None]
    if Lc < 8 or not np.all(is_self.sum(axis=1) == 1):
        return None
    order = np.argsort(is_self, axis=1, kind="stable")          # the four hops first, the site itself last
    nbr = np.take_along_axis(nbr, order, axis=1)
    blk = blk[site[:, None], order]
    hop, hop_to, diag = blk[:, :4], nbr[:, :4], blk[:, 4]
    if even[hop_to[even]].any() or not even[hop_to[~even]].all():
        return None                                                # not a nearest-neighbour operator
    E, O = site[even], site[~even]
    ginv = np.zeros((ns, 16, 16), dtype=np.complex128)
    ginv[O] = np.linalg.inv(diag[O])
    half = Lc // 2
    disp = [(2, 0), (-2, 0), (0, 2), (0, -2), (1, 1), (1, -1), (-1, 1), (-1, -1), (0, 0)]   # own site LAST
    lut = np.full((5, 5), -1)
    for q, (a, b) in enumerate(disp):
        lut[a + 2, b + 2] = q

    def even_chunk(Ec):
        """S and F block rows of the even sites Ec (packed), or None on an unexpected geometry."""
        F = hop[Ec] @ ginv[hop_to[Ec]]                              # [nc, 4]: A_eo D_oo^-1, onto hop_to[Ec]
        # S = D_ee - sum over the odd neighbours o of F(e, o) A(o, t): nine even targets t per e
        odd_nb = hop_to[Ec]                                         # [nc, 4]
        tgt = hop_to[odd_nb]                                        # [nc, 4, 4]
        prod = F[:, :, None] @ hop[odd_nb]                          # [nc, 4, 4, 16, 16]
        dx = (xs[tgt] - xs[Ec][:, None, None] + half) % Lc - half
        dy = (ys[tgt] - ys[Ec][:, None, None] + half) % Lc - half
        slot = lut[np.clip(dx, -2, 2) + 2, np.clip(dy, -2, 2) + 2]
        if (slot < 0).any() or (np.abs(dx) > 2).any() or (np.abs(dy) > 2).any():
            return None
        Sb = np.zeros((Ec.size, 9, 16, 16), dtype=np.complex128)
        Sb[:, 8] = diag[Ec]
        rows = np.arange(Ec.size)
        for j in range(4):
            for jp in range(4):
                Sb[rows, slot[:, j, jp]] -= prod[:, j, jp]          # one target per (e, j, jp): no clashes
        s_to = np.stack([((ys[Ec] + b) % Lc) * Lc + (xs[Ec] + a) % Lc for (a, b) in disp], axis=1)
        return Sb, s_to, pack_site_blocks(Sb, s_to), pack_site_blocks(F, hop_to[Ec])

    def odd_chunk(Oc):
        Hb = ginv[Oc][:, None] @ hop[Oc]                            # [nc, 4]: D_oo^-1 A_oe, onto hop_to[Oc]
        return pack_site_blocks(ginv[Oc][:, None], Oc[:, None]), pack_site_blocks(Hb, hop_to[Oc])

    # batched 16 x 16 products release the GIL: chunks of sites on a few host threads
    import concurrent.futures as cf
    import os
    nthr = max(1, min(16, (os.cpu_count() or 1)))
    nchunk = max(1, min(4 * nthr, E.size // 256))
    with cf.ThreadPoolExecutor(max_workers=nthr) as pool:
        ev = list(pool.map(even_chunk, np.array_split(E, nchunk)))
        od = list(pool.map(odd_chunk, np.array_split(O, nchunk)))
    if any(r is None for r in ev):
        return None
    ne = E.size
    Sb = np.concatenate([r[0] for r in ev])
    s_to = np.concatenate([r[1] for r in ev])
    cat = lambda parts: (np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts]))   # noqa: E731
    packed = [(E.astype(np.int32),) + cat([r[2] for r in ev]),
              (E.astype(np.int32),) + cat([r[3] for r in ev]),
              (O.astype(np.int32),) + cat([r[0] for r in od]),
              (O.astype(np.int32),) + cat([r[1] for r in od])]
    erank = np.full(ns, -1)
    erank[E] = np.arange(ne)
    S_ee = sp.bsr_matrix((Sb.reshape(-1, 16, 16), erank[s_to].reshape(-1), np.arange(0, 9 * ne + 1, 9)),
                         shape=(ne * 16, ne * 16))
    row_even = np.repeat(even, 16)
    return {"packed": packed, "S_ee": S_ee, "E_sites": E, "O_sites": O,
            "E_rows": np.nonzero(row_even)[0], "O_rows": np.nonzero(~row_even)[0]}


def dense_schur_inverse_blocks(ops):
    """(tmap, kcol, vals) of the DENSE inverse of a block level's even-odd Schur complement in the
    MFMA block-row form (row tiles = the even sites, every row tile lists all even sites' columns):
    even-odd operator 4 of the engine (sw_set_eo_operator), with which the level is solved exactly --
    x_e = S^-1 (b_e - F b_o), x_o = G b_o - Hb x_e -- instead of smoothed.  ops: coarse_schur_blocks()."""
    E = np.asarray(ops["E_sites"])
    ne = E.size
    Sinv = np.linalg.inv(ops["S_ee"].toarray())
    blocks = np.ascontiguousarray(Sinv.reshape(ne, 16, ne, 16).transpose(0, 2, 1, 3))
    targets = np.broadcast_to(E[None, :], (ne, ne))
    kcol, vals = pack_site_blocks(blocks, targets)
    return E.astype(np.int32), kcol, vals


def upload_coarse_eo(engines, hid, level, A_l, Lc, degree):
    """Build the four even-odd operators of block level `level` on the host, hand them to the engines
    and select `degree` Schur steps as its post-smoother.  A_l: the level operator as a sparse matrix,
    or its block-row form (kcol, vals) as sw_get_level_bsr returns it.  Returns (weights, ops)."""
    sb = site_blocks_from_block_rows(*A_l) if isinstance(A_l, tuple) else site_blocks_of(A_l, Lc)
    ops = coarse_schur_blocks(sb[0], sb[1], Lc) if sb is not None else None
    if ops is not None:
        weights = smoother_weights(ops["S_ee"], degree)
        for eng in engines:
            for which, (tmap, kcol, vals) in enumerate(ops["packed"]):
                eng.set_eo_operator(hid, level, which, tmap, kcol, vals)
            eng.set_eo_smoother(hid, level, weights)
        return weights, ops
    # general path (tiny or irregular lattices): sparse products and per-site packing
    if isinstance(A_l, tuple):
        A_l = matrix_from_block_rows(A_l[0], A_l[1], A_l[0].shape[0] * 16)
    ops = coarse_schur_operators(A_l, Lc)
    n = A_l.shape[0]
    packed = [block_rows_from_matrix(ops["S"], ops["E_sites"], n),
              block_rows_from_matrix(ops["F"], ops["E_sites"], n),
              block_rows_from_matrix(ops["G"], ops["O_sites"], n),
              block_rows_from_matrix(ops["Hb"], ops["O_sites"], n)]
    E = ops["E_rows"]
    weights = smoother_weights(sp.csr_matrix(ops["S"][E][:, E]), degree)
    for eng in engines:
        for which, (tmap, kcol, vals) in enumerate(packed):
            eng.set_eo_operator(hid, level, which, tmap, kcol, vals)
        eng.set_eo_smoother(hid, level, weights)
    return weights, ops


DEFAULT_SOLVER_CFG = {
    # (aggregate edge in sites of the level above, test vectors per chirality) per coarsening
    "coarsening": [(4, 8), (2, 8)],
    # per level: (nu_pre, nu_post, kcycle)
    "cycle": [(0, 7, 0), (0, 7, 0)],
    # "richardson": fixed-polynomial smoother (no inner products); "mr": adaptive MR steps
    "smoother": "richardson",
    "restart": 6,
    "eig_tol": 1.0e-6,
}


# The benchmark's hierarchy for schwinger128 (bench.py; any L with L/8 a multiple of 4), built entirely
# on the GPU (setup_gpu.py, 0.4 s): 8 x 8 site aggregates straight to a 4096-row level (16 x 16 sites x
# 16), then 2 x 2 to a 1024^2 dense inverse; both smoothed levels on their even-odd Schur complements.
# One MI355X, 3 streams, 10 outer iterations: 20.8k probe-samples/s (30k+ with precond_precision f32).
# History of the shape: 8.8k for DEFAULT_SOLVER_CFG (4096^2 dense inverse) -> 9.9k with a 16384-row
# level in between (4 x 4 aggregates first) -> 12.2k level 0 even-odd -> 16.7-18.1k every level
# even-odd -> 20.8k without the 16384-row level: with twelve Schur steps on the lattice level the
# coarser first coarse space is enough, and the level whose 16 x 16 blocks made up 20 % of the time is gone.
# -> 25.5k with the outer solve on the even-odd reduced system (engine option eo_solve: half-length Krylov
# vectors) -> 28k with nine Schur steps instead of twelve (11 iterations instead of 10: an outer
# iteration has become cheap enough for the optimum to move; profiles/r02_cfg_sweeps.txt).
TUNED_SOLVER_CFG_128 = {
    "coarsening": [(8, 8), (2, 8)],
    # eight Schur steps on the lattice level (round 3, with the 4096-row level solved directly and one
    # batch at a time per GPU: 7 / 8 / 9 / 10 / 11 steps -> 30.1k / 31.0k / 29.8k / 30.8k / 29.7k
    # probe-samples/s at 12 / 11 / 11 / 10 / 10 outer iterations, gpurun_out r03c); ten on the block level
    # where it is smoothed rather than solved (eo_direct = 0)
    "cycle": [(0, 8, 0), (0, 10, 0)],
    "smoother": "richardson",
    "eo_levels": [0, 1],        # levels smoothed on their even-odd Schur complement (half vectors)
    "restart": 4,               # the cycle is strong enough that short restarts keep the iteration count; Gram-form
                                # cycles, 2 / 3 / 4: 32.3k / 35.1k / 35.7k probe-samples/s in round 3; round 4, strict
                                # parity mode (12 iterations = three cycles of four): 34.6k -> 35.1k, at the
                                # reference's stopping point 36.9k -> 37.6k (profiles/r04_ab_sessions.txt, r04g).  The
                                # normal equations of a cycle of four are conditioned ~1e8: y to ~1e-8 relative,
                                # ample for a cycle that reduces the residual by ~1e-5 and always ends on the TRUE
                                # residual; a pivot below 1e-13 of its diagonal truncates the cycle (fg_gram_col)
    "setup": "device",
    # device setup: three RELAXATION sweeps per new level (setup_tol = 0: a fixed 32 unpreconditioned
    # GMRES(restart) steps each that damp the rough components of the random start vectors -- relaxation by
    # design, not a solve that runs into an iteration cap), then one pass of CONVERGED inverse iteration
    # preconditioned by the hierarchy itself (setup_refine; 2-3 iterations to 1e-2).  Replacing two of the
    # three relaxation sweeps by a second converged pass was measured worse (12 instead of 11 outer
    # iterations, gpurun_out r03b b_boot)
    "setup_sweeps": 3, "setup_tol": 0.0, "setup_maxiter": 32, "setup_refine": 1,
    # the 4096-row level is solved exactly in even-odd reduced form: dense inverse of its 2048-row Schur
    # complement (formed on the device, sw_setup_direct_level) on the matrix cores -- four launches per
    # visit instead of the ~16 of ten Schur steps plus the level below (round 3: 355 -> 223 launches per
    # batch, one stream 25.4k -> 28.7k probe-samples/s; the 1024-row level stays in the hierarchy for the
    # setup's coarse correction and for eo_direct = 0)
    "direct_levels": [1],
}
# cfg["eo_smoother"] = True: the post-smoothing steps of level 0 (cycle[0][1] of them) run on the
# even-odd Schur complement (sw_set_eo_smoother) instead of the full operator.


def synthetic_solver_cfg(L, nu0=10, setup="device", levels=None):
    """Solver hierarchy for a synthetic L x L lattice (BASELINE config 5; bench.py --workload synthetic
    and the full-size GPU test use the same one): 8 x 8 site aggregates once, then 2 x 2 until the
    coarsest level is 16 x 16 sites (4096 rows); every level smoothed on its even-odd Schur complement
    (operators built on the device), a 2-step K-cycle around the solve of level 1, plain V-cycle below.
    1024^2: levels 2097152 / 262144 / 65536 / 16384 / 4096 (profiles/r02_synthetic_lattices.txt).

    levels = 3: BASELINE config 5 AS WRITTEN ("3-level MG"): 8 x 8 site aggregates, then ONE coarsening of
    the block level straight to 16 x 16 sites (1024^2: 8 x 8 coarse sites per aggregate, blocks of 512 rows
    for the per-aggregate QR), levels 2097152 / 262144 / 4096.  The 262144-row level then has to carry what
    the three levels below it carry in the five-level hierarchy: more Schur steps and a longer K-cycle."""
    a0 = 8 if L % 8 == 0 and L // 8 >= 16 else 4
    base = {"smoother": "richardson", "restart": 3, "setup": setup,
            # (relaxation sweeps + one converged refinement pass: see TUNED_SOLVER_CFG_128)
            "setup_sweeps": 3, "setup_tol": 0.0, "setup_maxiter": 32, "setup_refine": 1}
    if levels == 3:
        Lc = L // a0
        if Lc % 16 or Lc // 16 > 8:
            raise Exception("no three-level hierarchy for a %d x %d lattice (coarse extent %d)" % (L, L, Lc))
        a1 = Lc // 16
        # Schur steps / K-cycle length of the 262144-row level at 1024^2, 128 probes (profiles/r04_ab_sessions.txt,
        # r04a/b): 12/3 212, 16/4 223, 16/6 222, 24/4 262, 32/2 260, 32/4 259 probe-samples/s (25 ... 15 outer
        # iterations): a plateau from 24 steps on
        nu1 = int(os.environ.get("SW_SYNTH3_NU1", "24"))
        k1 = int(os.environ.get("SW_SYNTH3_K1", "4"))
        return dict(base, coarsening=[[a0, 8], [a1, 8]], cycle=[[0, nu0, 0], [0, nu1, k1]],
                    eo_levels=[0, 1])
    if levels not in (None, 0):
        want = int(levels)
    else:
        want = None
    depth = [[a0, 8]]
    Lc = L // a0
    while Lc > 16 and Lc % 8 == 0 and (want is None or len(depth) < want - 1):
        depth.append([2, 8])
        Lc //= 2
    nsm = len(depth)
    # (8 Schur steps on level 1 as on the levels below: 564 against 550 probe-samples/s with 10 at 1024^2,
    # the same 9 iterations -- profiles/r03_ab_sessions.txt, r03am)
    cyc = [[0, nu0, 0]] + [[0, 8, 2 if i == 1 and i < nsm - 1 else 0] for i in range(1, nsm)]
    if nsm > 1:
        cyc[-1] = [0, 14, 0]
    return dict(base, coarsening=depth, cycle=cyc, eo_levels=list(range(nsm)))


def _site_prolongator(Al, Lf, hd, agg, nvec, tv, fine_level):
    """P for 2-D site aggregates with a chirality split.

    Rows: level 0 uses the reference order idx = half*V + site; coarse levels are ordered
    site-major, idx = (site*2 + half)*hd + k, so that the 2*hd dofs of a site are contiguous
    (one 16-row MFMA tile per site when 2*hd == 16).  Columns: (csite*2 + half)*nvec + v.
    Each (aggregate, half) block is orthonormalised by QR."""
    n = Al.shape[0]
    Lc = Lf // agg
    idx = np.arange(n)
    if fine_level:
        half = idx // (n // 2)
        site = idx % (n // 2)
    else:
        site = idx // (2 * hd)
        half = (idx // hd) % 2
    x, y = site % Lf, site // Lf
    block = ((y // agg) * Lc + (x // agg)) * 2 + half
    order = np.argsort(block, kind="stable")
    nblocks = 2 * Lc * Lc
    rows_per_block = n // nblocks
    rows_sorted = order.reshape(nblocks, rows_per_block)
    M = np.asarray(tv)[rows_sorted, :nvec]            # [blocks, rows, nvec]
    Q, _ = np.linalg.qr(M)
    rr = np.repeat(rows_sorted[:, :, None], nvec, axis=2)
    cc = (np.arange(nblocks)[:, None, None] * nvec + np.arange(nvec)[None, None, :]
          + np.zeros((1, rows_per_block, 1), dtype=np.int64))
    return sp.csr_matrix((Q.ravel(), (rr.ravel(), cc.ravel())), shape=(n, nblocks * nvec))


def solver_hierarchy(A0, L, cfg=None, testvectors=None):
    """Level-0 preconditioner hierarchy: returns dict(A=[...], P=[...], coarsest_inv, tv)."""
    cfg = dict(DEFAULT_SOLVER_CFG if cfg is None else cfg)
    As = [sp.csr_matrix(A0).astype(np.complex128)]
    Ps = []
    tvs = []
    Lf, hd = L, 1
    for lvl, (agg, nvec) in enumerate(cfg["coarsening"]):
        if Lf % agg:
            raise Exception("lattice extent %d not divisible by aggregate edge %d" % (Lf, agg))
        if testvectors is not None:
            tv = testvectors[lvl]
        else:
            _, tv = spla.eigs(sp.csc_matrix(As[-1]), k=nvec, which="LM", sigma=0.0,
                              tol=cfg.get("eig_tol", 1e-6))
        tvs.append(tv)
        P = _site_prolongator(As[-1], Lf, hd, agg, nvec, tv, lvl == 0)
        Ac = sp.csr_matrix(P.conjugate().transpose() @ As[-1] @ P)
        Ps.append(P)
        As.append(Ac)
        Lf //= agg
        hd = nvec
    cinv = dense_inverse(As[-1].toarray())
    return {"A": As, "P": Ps, "coarsest_inv": cinv, "tv": tvs, "cfg": cfg}
