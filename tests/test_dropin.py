"""CPU: the reference's ``main.py`` import lines resolve against the drop-in aliases."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_style_imports_resolve():
    code = ("from gateway import G101,G102\nfrom gateway import G201,G202\n"
            "from examples import EXAMPLE_001, EXAMPLE_002\nfrom matrix import loadMatrix\n"
            "from stoch_trace import hutchinson,mlmc\n"
            "from utils import print_post_results,trace_params_from_params,CustomTimer,"
            "flopsV_manual,deflation_pre_computations,one_defl_Hutch_step\n"
            "from multigrid import MG, LevelML, SimpleML\n"
            "import gateway; p = gateway.set_params('schwinger128'); print(p['matrix'])\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         cwd="/tmp", timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "schwinger128.mat"
