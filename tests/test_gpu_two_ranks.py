"""GPU: the N > 1 path end to end on the one GPU of the test box -- two ranks (processes) share
the card, `torch.distributed` over gloo carries the per-round gather, every rank generates only
its own contiguous block of each round's probes on the device.  The drop-in hutchinson() / mlmc()
must give what one rank gives: same stopping index, same stream position on return, estimates
equal to solver accuracy (batches are composed differently, so the lockstep iteration counts and
hence the last digits of converged solves may differ)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = {"function_tol": 1e-12, "nr_deflat_vctrs": 8, "mlmc_deflat_vctrs": [0, 0],
          "trace_tol": 3.0e-2, "batch": 16, "accuracy_mg_eigvs": "high"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    worker = os.path.join(HERE, "two_rank_worker.py")
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    one = str(tmp_path / "one")
    two = str(tmp_path / "two")
    env1 = dict(env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env1.pop(k, None)
    r = subprocess.run([sys.executable, worker, one, json.dumps(PARAMS)], env=env1,
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), worker, two, json.dumps(PARAMS)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = json.load(open(one + ".rank0"))
    for rank in (0, 1):
        got = json.load(open(two + ".rank%d" % rank))
        h, hr = got["hutchinson"], ref["hutchinson"]
        assert h["nr_ests"] == hr["nr_ests"] and h["next_draw"] == hr["next_draw"]
        e, er = np.array(h["ests"]), np.array(hr["ests"])
        assert e.shape == er.shape
        assert np.max(np.abs(e - er)) < 1e-8 * np.max(np.abs(er))
        assert abs(h["std_dev"] - hr["std_dev"]) < 1e-8 * hr["std_dev"]
        m, mr = got["mlmc"], ref["mlmc"]
        assert m["next_draw"] == mr["next_draw"]
        for lv, lr in zip(m["levels"], mr["levels"]):
            assert lv["nr_ests"] == lr["nr_ests"]
            assert abs(lv["ests_dev"] - lr["ests_dev"]) <= 1e-7 * max(1.0, lr["ests_dev"])
            assert np.max(np.abs(np.array(lv["ests_avg"]) - np.array(lr["ests_avg"]))) < 1e-7 * max(
                1.0, np.max(np.abs(lr["ests_avg"])))
    # both ranks of the 2-rank run agree with each other to the last bit (same gathered values)
    a, b = (json.load(open(two + ".rank%d" % k)) for k in (0, 1))
    assert a == b
