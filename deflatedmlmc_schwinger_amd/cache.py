"""Optional on-disk cache of the expensive host-side setup products: the ARPACK
test vectors of the reference hierarchy, those of the solver hierarchy and the deflation
eigenpairs, keyed by a hash of the matrix and of the parameters that determine them.  Enabled by
``params['cache_dir']`` or the environment variable ``SW_CACHE_DIR``; everything cheap (Galerkin
products, per-aggregate orthonormalisation, dense inverse) is always recomputed.

Also: ``write_run_report`` appends one JSON line per estimator run (trace, error, samples/s).
"""
import hashlib
import json
import os
import time

import numpy as np
import scipy.sparse as sp


def cache_dir(params):
    d = None
    if params is not None and hasattr(params, "get"):
        d = params.get("cache_dir")
    d = d or os.environ.get("SW_CACHE_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
    return d


def matrix_key(A, extra):
    A = sp.csr_matrix(A)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(A.indptr).tobytes())
    h.update(np.ascontiguousarray(A.indices).tobytes())
    h.update(np.ascontiguousarray(A.data).tobytes())
    h.update(json.dumps(extra, sort_keys=True, default=str).encode())
    return h.hexdigest()[:24]


def load(directory, name, key):
    if not directory:
        return None
    path = os.path.join(directory, "%s_%s.npz" % (name, key))
    if not os.path.exists(path):
        return None
    with np.load(path, allow_pickle=False) as z:
        return {k: np.array(z[k]) for k in z.files}


def save(directory, name, key, arrays):
    if not directory:
        return
    path = os.path.join(directory, "%s_%s.npz" % (name, key))
    tmp = "%s.tmp.%d.npz" % (path, os.getpid())
    np.savez(tmp, **arrays)
    os.replace(tmp, path)          # atomic: several ranks may write the same entry


def write_run_report(result, kind, params, elapsed_s, extra=None):
    """One JSON line per run into $SW_REPORT_PATH (or params['report_path']) if set."""
    path = (params.get("report_path") if hasattr(params, "get") else None) or \
        os.environ.get("SW_REPORT_PATH")
    if not path:
        return None
    rec = {"time": time.time(), "kind": kind, "matrix": params.get("matrix"),
           "trace": [float(np.real(result["trace"])), float(np.imag(result["trace"]))],
           "elapsed_s": elapsed_s}
    if kind == "hutchinson":
        n = int(result["nr_ests"]) + 1
        rec.update(std_dev=float(result["std_dev"]), nr_ests=int(result["nr_ests"]),
                   function_iters=int(result["function_iters"]),
                   probe_samples_per_s=n / elapsed_s if elapsed_s > 0 else None)
        if result.get("probe_loop_s"):
            # the probe loop alone (setup, deflation vectors and the rough trace excluded): probes the
            # GPU solved per second, and the ones the stopping rule ended up using
            rec.update(probe_loop_s=float(result["probe_loop_s"]),
                       probes_solved=int(result["probes_solved"]),
                       probe_loop_solved_per_s=result["probes_solved"] / result["probe_loop_s"],
                       probe_loop_used_per_s=n / result["probe_loop_s"])
    else:
        rec["levels"] = [{"nr_ests": int(r["nr_ests"]), "function_iters": int(r["function_iters"]),
                          "ests_avg": [float(np.real(r["ests_avg"])), float(np.imag(r["ests_avg"]))],
                          "ests_dev": float(r["ests_dev"]),
                          "probe_loop_s": float(r.get("probe_loop_s", 0.0)),
                          "probes_solved": int(r.get("probes_solved", 0))} for r in result["results"]]
        solved = sum(int(r.get("probes_solved", 0)) for r in result["results"])
        loop_s = sum(float(r.get("probe_loop_s", 0.0)) for r in result["results"])
        if loop_s > 0:
            rec.update(probes_solved=solved, probe_loop_s=loop_s, probe_loop_solved_per_s=solved / loop_s)
    if extra:
        rec.update(extra)
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")
    return rec
