"""NumPy model of the ENGINE's own multigrid cycle (MR(nu)-smoothed V/K-cycle, batched
flexible GMRES).  TEST INFRASTRUCTURE ONLY: it lets the GPU tests compare the HIP cycle
against an independent CPU evaluation of the same formulae on the same operands.  The
algorithm itself is the build's, not the reference's; the reference's
cycle (lgmres smoother) lives in oracle/ref_path.py."""
import numpy as np


def cdot(X, Y):
    return np.einsum("ij,ij->j", X.conj(), Y)


def mr_smooth(A, X, R, nu):
    for _ in range(nu):
        T = A @ R
        num = cdot(T, R)
        den = cdot(T, T).real
        alpha = np.where(den > 0, num / np.where(den > 0, den, 1.0), 0.0)
        X = X + R * alpha
        R = R - T * alpha
    return X, R


def fgmres_fixed(A, B, M, k):
    """k steps of right-preconditioned flexible GMRES from a zero guess, per column."""
    n, nb = B.shape
    beta = np.sqrt(cdot(B, B).real)
    V = [B * np.where(beta > 0, 1.0 / np.where(beta > 0, beta, 1.0), 0.0)]
    Z = []
    H = np.zeros((k + 1, k, nb), dtype=complex)
    for j in range(k):
        Zj = M(V[j])
        W = A @ Zj
        Z.append(Zj)
        for _ in range(2):
            for i in range(j + 1):
                h = cdot(V[i], W)
                H[i, j] += h
                W = W - V[i] * h
        hn = np.sqrt(cdot(W, W).real)
        H[j + 1, j] = hn
        V.append(W * np.where(hn > 0, 1.0 / np.where(hn > 0, hn, 1.0), 0.0))
    X = np.zeros_like(B)
    for c in range(nb):
        e1 = np.zeros(k + 1, dtype=complex)
        e1[0] = beta[c]
        y = np.linalg.lstsq(H[:, :, c], e1, rcond=None)[0]
        for i in range(k):
            X[:, c] += y[i] * Z[i][:, c]
    return X


def richardson(A, B, X, weights, from_zero):
    """x <- x + w_k (B - A x) for the given complex weights."""
    for k, w in enumerate(weights):
        if from_zero and k == 0:
            X = w * B
        else:
            X = X + w * (B - A @ X)
    return X


def eo_post_smooth(A, B, X, w_eo, E, O):
    """Even-odd post-smoothing of one level (sw_set_eo_smoother): x_e <- x_e + w (b'_e - S x_e) on the
    rows E with S = A_ee - A_eo A_oo^-1 A_oe, b'_e = b_e - A_eo A_oo^-1 b_o, then the rows O exactly.
    A_oo is (block) diagonal; the model inverts it densely per solve (small test operands)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    A = sp.csr_matrix(A)
    Aee, Aeo = A[E][:, E], A[E][:, O]
    Aoe, Aoo = A[O][:, E], sp.csc_matrix(A[O][:, O])
    lu = spla.splu(Aoo)
    xe = X[E]
    bp = B[E] - Aeo @ lu.solve(B[O])
    for w in w_eo:
        Sx = Aee @ xe - Aeo @ lu.solve(Aoe @ xe)
        xe = xe + w * (bp - Sx)
    out = np.zeros_like(B)
    out[E] = xe
    out[O] = lu.solve(B[O] - Aoe @ xe)
    return out


def cycle(As, Ps, cinv, cfg, level, B, weights=None, eo=None):
    """cfg[l] = (nu_pre, nu_post, kcycle); B is [n, nb].  weights[l] = (w_pre, w_post) selects the
    fixed-polynomial smoother at level l (None: adaptive MR steps); eo[l] = (w_eo, E, O) replaces the
    post-smoother of level l by the even-odd Schur smoother."""
    last = len(As) - 1
    if level == last:
        return np.asarray(cinv) @ B
    nu_pre, nu_post, kc = cfg[level]
    A, P = As[level], Ps[level]
    R = P.conj().T
    if weights is not None and weights[level] is not None:
        w_pre, w_post = weights[level]
        use_eo = eo is not None and level in eo
        if len(w_pre) and not use_eo:
            X = richardson(A, B, None, w_pre, True)
            Bc = R @ (B - A @ X)
        else:
            X = None
            Bc = R @ B
        sub = lambda v: cycle(As, Ps, cinv, cfg, level + 1, v, weights, eo)   # noqa: E731
        Xc = fgmres_fixed(As[level + 1], Bc, sub, kc) if (kc > 0 and level + 1 < last) else sub(Bc)
        X = P @ Xc if X is None else X + P @ Xc
        if use_eo:
            return eo_post_smooth(A, B, X, *eo[level])
        return richardson(A, B, X, w_post, False)
    if nu_pre > 0:
        X, res = mr_smooth(A, np.zeros_like(B), B.copy(), nu_pre)
        Bc = R @ res
    else:
        X = None
        Bc = R @ B
    if kc > 0 and level + 1 < last:
        Xc = fgmres_fixed(As[level + 1], Bc, lambda v: cycle(As, Ps, cinv, cfg, level + 1, v), kc)
    else:
        Xc = cycle(As, Ps, cinv, cfg, level + 1, Bc)
    X = P @ Xc if X is None else X + P @ Xc
    if nu_post > 0:
        X, _ = mr_smooth(A, X, B - A @ X, nu_post)
    return X


def cycle_eo(As, Ps, cinv, cfg, B, weights, w_eo, E, O, D):
    """The level-0 cycle with the even-odd post-smoother (sw_set_eo_smoother): coarse correction as
    in cycle(), then w_eo Richardson steps on S = D - A_eo A_oe / D for the even sites E and the odd
    sites O solved exactly.  B is [n, nb] in the reference order."""
    A, P = As[0], Ps[0]
    R = P.conj().T
    sub = lambda v: cycle(As, Ps, cinv, cfg, 1, v, weights)   # noqa: E731
    X = P @ sub(R @ B)
    Aeo = A[E][:, O]
    Aoe = A[O][:, E]
    xe = X[E]
    bp = B[E] - Aeo @ (B[O] / D)
    for w in w_eo:
        Sx = D * xe - Aeo @ (Aoe @ xe) / D
        xe = xe + w * (bp - Sx)
    out = np.zeros_like(B)
    out[E] = xe
    out[O] = (B[O] - Aoe @ xe) / D
    return out


def fgmres_restarted(op, B, M, tol, m, maxiter, normb=None):
    """Restarted right-preconditioned flexible GMRES(m) from a zero guess, all columns in lockstep (two
    Gram-Schmidt passes); stops when every column's TRUE residual is below tol * normb.  Returns
    (X, iterations).  `op`, `M`: callables on [n, nb] arrays."""
    nb = B.shape[1]
    normb = np.sqrt(cdot(B, B).real) if normb is None else normb
    X = np.zeros_like(B)
    its = 0
    while its < maxiter:
        R = B - op(X)
        if np.all(np.sqrt(cdot(R, R).real) < tol * normb):
            break
        beta = np.sqrt(cdot(R, R).real)
        V = [R / np.where(beta > 0, beta, 1.0)]
        Z = []
        k = min(m, maxiter - its)
        H = np.zeros((k + 1, k, nb), dtype=complex)
        for j in range(k):
            Z.append(M(V[j]))
            W = op(Z[j])
            for _ in range(2):
                for i in range(j + 1):
                    h = cdot(V[i], W)
                    H[i, j] += h
                    W = W - V[i] * h
            hn = np.sqrt(cdot(W, W).real)
            H[j + 1, j] = hn
            V.append(W / np.where(hn > 0, hn, 1.0))
        for c in range(nb):
            e1 = np.zeros(k + 1, dtype=complex)
            e1[0] = beta[c]
            y = np.linalg.lstsq(H[:, :, c], e1, rcond=None)[0]
            for i in range(k):
                X[:, c] += y[i] * Z[i][:, c]
        its += k
    return X, its


def solve_even_odd_reduced(A, B, M, E, O, D, tol, m, maxiter):
    """The engine's fgmres_eo in NumPy: A x = b through the even-odd reduced system
       S x_e = b'_e,  b'_e = b_e - A_eo b_o / D,  x_o = (b_o - A_oe x_e) / D,  S = D - A_eo A_oe / D,
    with half-length Krylov vectors, the preconditioner M_S r_e = [M (r_e; 0)]_e (M: the full-lattice
    cycle) and the residual measured against ||b|| of the FULL system.  Returns (X, iterations)."""
    Aeo, Aoe = A[E][:, O], A[O][:, E]
    n = A.shape[0]

    def S(xe):
        return D * xe - Aeo @ (Aoe @ xe) / D

    def MS(re):
        full = np.zeros((n, re.shape[1]), dtype=complex)
        full[E] = re
        return M(full)[E]

    bp = B[E] - Aeo @ (B[O] / D)
    xe, its = fgmres_restarted(S, bp, MS, tol, m, maxiter, normb=np.sqrt(cdot(B, B).real))
    X = np.zeros_like(B)
    X[E] = xe
    X[O] = (B[O] - Aoe @ xe) / D
    return X, its
