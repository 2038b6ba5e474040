#!/usr/bin/env python3
"""Exploratory: synthetic L x L lattice (BASELINE config 5), GPU-side adaptive setup, one batch of
plain Hutchinson probes; prints setup time, iterations, probes/s and the true residual check."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
from deflatedmlmc_schwinger_amd import hierarchy, matrix
from deflatedmlmc_schwinger_amd.engine import MODE_HUTCHINSON, ProbeStream
from deflatedmlmc_schwinger_amd.multigrid import MG, SOLVER_HID

L = int(sys.argv[1]); nb = int(sys.argv[2]); cfg = json.loads(sys.argv[3])
mass = float(os.environ.get("SW_MASS", "-0.02")); sigma = float(os.environ.get("SW_SIGMA", "0.35"))
t0 = time.time()
U1, U2 = matrix.synthetic_links(L, sigma, 2024)
lat = (L, mass, U1, U2)
mg = MG(lat)
mg.setup_solver_only(cfg)
t_setup = time.time() - t0
eng = mg.engine
n = 2 * L * L
probes = ProbeStream(123456).rademacher(nb, n)
eng.probes_upload(0, probes)
eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000); eng.sync()
t0 = time.perf_counter(); eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000); eng.sync(); dt = time.perf_counter() - t0
ests, itf, _ = eng.hutch_fetch()
# true residual of a fresh solve on 2 probes
B = probes[:2].astype(np.complex128)
X, its, rr = eng.solve(SOLVER_HID, 0, B, 1e-12, 1000)
AX = eng.apply_dirac(0, 0, X)
true_rel = np.linalg.norm(B - AX, axis=1) / np.linalg.norm(B, axis=1)
eng.set_profiling(True); eng.timers_reset(); eng.hutch_run(MODE_HUTCHINSON, 0, 1e-12, 1000); b = eng.timers(); eng.set_profiling(False)
print(json.dumps({"L": L, "nb": nb, "levels": mg.solver_info["levels"], "setup_s": round(t_setup, 1),
                  "setup_log": mg.solver_info["setup_log"], "iters": int(itf.max()), "ms": round(1e3 * dt, 1),
                  "probes_per_s": round(nb / dt, 1), "true_relres": float(true_rel.max()),
                  "buckets_ms": {k: round(v, 1) for k, v in b.items()}, "e0": [ests[0].real, ests[0].imag]}))
