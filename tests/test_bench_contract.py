"""CPU: bench.py keeps its command-line contract and refuses to run without a GPU (no CPU path)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags_and_no_gpu_exit():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--streams", "--workload"):
        assert flag in out.stdout
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "no CPU path" in (out.stderr + out.stdout)
    assert out.stdout.strip() == ""          # nothing but the JSON line ever goes to stdout
