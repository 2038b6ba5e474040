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


def test_probe_blocks_partition_the_stream_in_single_gpu_order():
    """bench.first_probe (BASELINE config 4's sharding): over all (rank, stream) of a round the blocks are
    disjoint, contiguous and in stream order, rounds follow each other without gaps, and the union over
    ranks is what one GPU with world * streams engines would have drawn."""
    sys.path.insert(0, ROOT)
    import bench
    for world, streams, nb in ((1, 3, 256), (8, 3, 256), (4, 2, 64), (2, 1, 100)):
        starts = []
        for step in range(3):
            for rank in range(world):
                for e in range(streams):
                    starts.append(bench.first_probe(step, world, rank, streams, e, nb))
        assert starts == [k * nb for k in range(3 * world * streams)]
        one_gpu = [bench.first_probe(step, 1, 0, world * streams, e, nb)
                   for step in range(3) for e in range(world * streams)]
        assert one_gpu == starts
