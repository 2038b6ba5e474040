"""``matrix.loadMatrix`` of the reference (matrix.py:14-31) plus the U(1)-link view of the
same data that the matrix-free GPU stencil consumes.

``loadMatrix(matrix_name, params)`` keeps the reference behaviour: read variable ``S`` from
the MATLAB file, undo the gamma3 flip for ``schwinger16.mat`` only, add ``mass`` on the
diagonal, return a SciPy sparse matrix.  When the ``.mat`` file is not present (the two
gauge configurations are the reference's data files and do not ship with this repository)
the same matrix is rebuilt, bit for bit, from the link fixture under ``data/``
(``<name>.links.npz``, produced from the ``.mat`` by ``tests/golden/make_golden.py``).
"""
import os
import warnings

import numpy as np
from scipy.sparse import csr_matrix, identity, diags

from .hierarchy import links_from_matrix, wilson_from_links

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def links_fixture_path(matrix_name):
    base = os.path.splitext(os.path.basename(matrix_name))[0]
    return os.path.join(_DATA_DIR, base + ".links.npz")


def load_links(matrix_name):
    """(L, U1, U2) of a shipped gauge configuration."""
    path = links_fixture_path(matrix_name)
    if not os.path.exists(path):
        raise FileNotFoundError("neither %s nor the link fixture %s exists" % (matrix_name, path))
    with np.load(path) as z:
        return int(z["L"]), np.array(z["U1"]), np.array(z["U2"])


def load_S(matrix_name):
    """The hopping matrix S (diagonal 4) in the reference's index order."""
    if os.path.exists(matrix_name):
        import scipy.io as sio
        S = csr_matrix(sio.loadmat(matrix_name)['S']).astype(np.complex128)
        # matrix.py:24-27: the 16^2 file stores gamma3*D
        if os.path.basename(matrix_name) == 'schwinger16.mat' or matrix_name == 'schwinger16.mat':
            half = int(S.shape[0] / 2)
            sign = np.ones(S.shape[0])
            sign[half:] = -1.0
            S = csr_matrix(diags(sign) @ S)
        return S
    L, U1, U2 = load_links(matrix_name)
    return wilson_from_links(U1, U2, L)


def loadMatrix(matrix_name, params):
    warnings.simplefilter("ignore")
    m = params['mass']
    A = load_S(matrix_name)
    A = A + m * identity(A.shape[0], dtype=A.dtype)
    return csr_matrix(A)


def save_links_fixture(mat_path, out_path=None):
    """Extract the U(1) links of a reference ``.mat`` file into ``data/<name>.links.npz``."""
    S = load_S(mat_path)
    L = int(round(np.sqrt(S.shape[0] // 2)))
    U1, U2 = links_from_matrix(S, L)
    rebuilt = wilson_from_links(U1, U2, L)
    d = abs(S - rebuilt)
    if d.nnz and d.max() != 0.0:
        raise Exception("link extraction does not reproduce the matrix exactly")
    out_path = out_path or links_fixture_path(mat_path)
    np.savez_compressed(out_path, L=L, U1=U1, U2=U2)
    return out_path


def synthetic_links(L, sigma=0.45, seed=2024):
    """Random U(1) gauge field for the synthetic configurations of BASELINE config 5:
    U_mu(n) = exp(i theta), theta ~ N(0, sigma^2) i.i.d. (sigma ~ 0.41 gives a mean plaquette of
    about 0.92 / e^{-2 sigma^2}-like decorrelation comparable to schwinger128)."""
    rng = np.random.default_rng(seed)
    th = rng.normal(0.0, sigma, size=(2, L * L))
    return np.exp(1j * th[0]), np.exp(1j * th[1])


def synthetic_matrix(L, mass, sigma=0.45, seed=2024):
    """A = S + mass*I for a synthetic L x L configuration, reference index order."""
    U1, U2 = synthetic_links(L, sigma, seed)
    S = wilson_from_links(U1, U2, L)
    return csr_matrix(S + mass * identity(S.shape[0], dtype=S.dtype))
