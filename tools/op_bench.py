#!/usr/bin/env python3
"""Back-to-back timing of ONE operator kernel of the schwinger128 hierarchies (for A/B runs and for
rocprofv3 --pmc passes on a single kernel): python tools/op_bench.py --hid 1 --level 1 --mode 2
[--what op|R|P|cinv] [--opt name=value ...].  Test vectors are cached under --cache so repeated
invocations inside one gpurun call skip ARPACK."""
import argparse
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hid", type=int, default=1)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--mode", type=int, default=2)
    ap.add_argument("--what", default="op", choices=["op", "R", "P", "cinv"])
    ap.add_argument("--nb", type=int, default=256)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--cache", default="/tmp/swcache")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    from deflatedmlmc_schwinger_amd import gateway, matrix, utils
    from deflatedmlmc_schwinger_amd.multigrid import MG
    params = gateway.set_params('schwinger128')
    params['function_tol'] = 1e-12
    params['cache_dir'] = args.cache
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "hutchinson")
    mg = MG(A)
    with contextlib.redirect_stdout(io.StringIO()):
        mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
                 acc_eigvs=tp['accuracy_mg_eigvs'], sys_type='schwinger', params=tp)
    eng = mg.engine
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, float(v))
    eng.set_option("bench_mode", args.mode)
    eng.set_option("bench_what", {"op": 0, "R": 1, "P": 2, "cinv": 3}[args.what])
    ms = eng.bench_dirac(args.hid, args.level, args.nb, args.reps)
    print(json.dumps({"hid": args.hid, "level": args.level, "mode": args.mode, "what": args.what,
                      "opts": args.opt, "ms": ms}))


if __name__ == "__main__":
    main()
