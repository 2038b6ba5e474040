#!/usr/bin/env python3
"""GPU tuning aid: time one 256-probe deflated-Hutchinson batch for several solver-hierarchy
configurations (hierarchy shape, smoothing steps, K-cycle depth, restart length)."""
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

from deflatedmlmc_schwinger_amd import gateway, matrix, utils  # noqa: E402
from deflatedmlmc_schwinger_amd.engine import MODE_HUTCHINSON, ProbeStream  # noqa: E402
from deflatedmlmc_schwinger_amd.multigrid import MG  # noqa: E402


def main():
    cfgs = json.load(open(sys.argv[1])) if len(sys.argv) > 1 else []
    nb = int(os.environ.get("SW_NB", "256"))
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "hutchinson")
    mg = MG(A)
    with contextlib.redirect_stdout(io.StringIO()):
        mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
                 acc_eigvs=tp['accuracy_mg_eigvs'], sys_type=tp['problem_name'], params=tp)
        utils.deflation_pre_computations(A, 8, 1e-9, "hutchinson", mg.timer, tp, mg)
    eng = mg.engine
    probes = ProbeStream(123456).rademacher(nb, A.shape[0])
    eng.probes_upload(0, probes)
    last_key, tv = None, None
    for cfg in cfgs:
        key = json.dumps([cfg["coarsening"], cfg.get("setup"), cfg.get("setup_sweeps"), cfg.get("setup_tol"), cfg.get("setup_maxiter")])
        t0 = time.time()
        try:
            mg.upload_solver_hierarchy(cfg, testvectors=tv if key == last_key else None)
        except Exception as e:
            print(json.dumps({"cfg": cfg, "error": repr(e)}), flush=True)
            continue
        last_key, tv = key, mg.solver_testvectors
        eng.set_option("use_mfma", float(cfg.get("use_mfma", 1)))
        eng.set_option("mfma_tiles", float(cfg.get("mfma_tiles", 4)))
        for key, val in cfg.items():
            if key.startswith("opt_"):
                eng.set_option(key[4:], float(val))
        t_setup = time.time() - t0
        eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000)
        eng.sync()
        t0 = time.perf_counter()
        eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000)
        eng.sync()
        dt = time.perf_counter() - t0
        ests, itf, _ = eng.hutch_fetch()
        eng.set_profiling(True)
        eng.timers_reset()
        eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000)
        b = eng.timers()
        launches = eng.launch_count()
        eng.set_profiling(False)
        print(json.dumps({"cfg": cfg, "levels": mg.solver_info["levels"], "ms": 1e3 * dt,
                          "setup_log": mg.solver_info.get("setup_log"),
                          "probes_per_s": nb / dt, "iters": int(itf.max()), "setup_s": t_setup,
                          "buckets_ms": {k: round(v, 2) for k, v in b.items()},
                          "launches": launches, "e0": [ests[0].real, ests[0].imag]}), flush=True)


if __name__ == "__main__":
    main()
