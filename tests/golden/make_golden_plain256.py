#!/usr/bin/env python3
"""256 plain (k = 0) Hutchinson probes on schwinger128 from the REFERENCE's own probe body
(utils.one_defl_Hutch_step with exact sparse-LU solves, as tests/golden/make_golden.py): the
full first batch of BASELINE config 2, seed 123456 -- SURVEY 8c asks for >= 256 golden 128^2 probes.
Hierarchy- and eigenvector-independent.  Run in the build container only (needs /root/reference);
writes tests/golden/hutch128_plain256.json."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mk          # noqa: E402  (imports the reference's utils / matrix)

ref_utils, ref_matrix, rp = mk.ref_utils, mk.ref_matrix, mk.rp


def main():
    mass128 = -0.1320
    cwd = os.getcwd()
    os.chdir(mk.REF)
    with mk.silence():
        A128 = sp.csr_matrix(ref_matrix.loadMatrix('schwinger128.mat',
                                                   {'mass': mass128, 'problem_name': 'schwinger'}))
    os.chdir(cwd)
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
    duck = mk.DuckSolver([lev0], np.eye(2), False)
    np.random.seed(123456)
    plain = []
    with mk.silence():
        for _ in range(256):
            e, _ = ref_utils.one_defl_Hutch_step(A128, None, duck, params128, "hutchinson", 0, None, None)
            plain.append(e)
    out = {"hutch128_plain_seed123456_256": mk.cplx_list(plain),
           "note": "reference utils.one_defl_Hutch_step (utils.py:210-250), k = 0, Pperm shift 512, exact "
                   "sparse-LU solves; probes 0..255 of np.random.seed(123456)"}
    with open(os.path.join(HERE, "hutch128_plain256.json"), "w") as f:
        json.dump(out, f)
    print("wrote 256 probes; first:", plain[0])


if __name__ == "__main__":
    main()
