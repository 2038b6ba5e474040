"""Example drivers with the reference's names (examples.py:13-51)."""
import time

from .cache import write_run_report
from .matrix import loadMatrix
from .stoch_trace import hutchinson, mlmc
from .utils import print_post_results, trace_params_from_params


def _banner(text):
    bar = "-" * len(text)
    print("\n" + bar)
    print(text)
    print(bar + "\n")


def _run(params, kind):
    A = loadMatrix(params['matrix'], params['matrix_params'])
    trace_params = trace_params_from_params(params, kind)
    estimator = hutchinson if kind == "hutchinson" else mlmc
    t0 = time.time()
    result = estimator(A, trace_params)
    elapsed = time.time() - t0
    write_run_report(result, kind, params, elapsed)
    return A, result, elapsed


# deflated Hutchinson                                                  examples.py:13-29
def EXAMPLE_001(params):
    _banner("Example 01 : computing tr(A^{-1}) with deflated Hutchinson")
    A, result, elapsed = _run(params, "hutchinson")
    print("Total Hutchinson time = " + str(elapsed) + " cpu seconds\n")
    print_post_results(A, params, result, "hutchinson")
    return result


# multigrid multilevel Monte Carlo                                     examples.py:35-51
def EXAMPLE_002(params):
    _banner("Example 02 : computing tr(A^{-1}) with MLMC")
    A, result, elapsed = _run(params, "mlmc")
    print("Total MLMC time = " + str(elapsed) + " cpu seconds")
    print_post_results(A, params, result, "mlmc")
    return result
