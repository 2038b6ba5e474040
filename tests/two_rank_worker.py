"""Worker of tests/test_gpu_two_ranks.py: one rank of a 2-rank run that shares ONE GPU (gloo
transport).  Runs the drop-in hutchinson() and mlmc() on schwinger16 and writes its results."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402


def run(out_path, params_over):
    import contextlib
    import io
    import torch.distributed as td
    from deflatedmlmc_schwinger_amd import gateway, matrix, stoch_trace, utils
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        td.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for example in ("hutchinson", "mlmc"):
        params = gateway.set_params('schwinger16')
        params.update(params_over)
        A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
        tp = utils.trace_params_from_params(params, example)
        with contextlib.redirect_stdout(io.StringIO()):
            r = (stoch_trace.hutchinson if example == "hutchinson" else stoch_trace.mlmc)(A, tp)
        if example == "hutchinson":
            res[example] = {"trace": [r['trace'].real, r['trace'].imag], "std_dev": r['std_dev'],
                            "nr_ests": int(r['nr_ests']), "function_iters": int(r['function_iters']),
                            "ests": [[e.real, e.imag] for e in r['ests']]}
        else:
            res[example] = {"trace": [complex(r['trace']).real, complex(r['trace']).imag],
                            "levels": [{"nr_ests": int(l['nr_ests']),
                                        "ests_avg": [complex(l['ests_avg']).real, complex(l['ests_avg']).imag],
                                        "ests_dev": float(l['ests_dev'])} for l in r['results']]}
        res[example]["next_draw"] = int(np.random.randint(1 << 30))
    with open(out_path + ".rank%d" % rank, "w") as f:
        json.dump(res, f)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1], json.loads(sys.argv[2]))
