#!/bin/bash
# round 3: synthetic 1024^2 lattice with wider batches (time-skewed strips walked 64-probe chunk by chunk)
OUT=gpurun_out/${1:-r03aa}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "product_form or time_skewed" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for nb in 64 128 256; do
  timeout -k 10 400 python bench.py --workload synthetic --lattice 1024 --nb $nb --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs > $OUT/b1024_nb$nb.json 2> $OUT/b1024_nb$nb.err || { tail -5 $OUT/b1024_nb$nb.err; exit 1; }
done
timeout -k 10 400 python bench.py --workload synthetic --lattice 512 --nb 256 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs > $OUT/b512_nb256.json 2> $OUT/b512_nb256.err
timeout -k 10 400 python bench.py --workload synthetic --lattice 512 --nb 64 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs > $OUT/b512_nb64.json 2> $OUT/b512_nb64.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    r = d.get("roofline") or {}
    print("%-22s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f P %.2f R %.2f dots %.2f axpy %.2f coarsest %.2f other %.2f | %s avg %.2f us frac %.3f setup %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), sb.get("P", 0), sb.get("R", 0), sb.get("dots", 0), sb.get("axpy", 0), sb.get("coarsest", 0), sb.get("other", 0),
        r.get("kernel"), 1e3 * (r.get("avg_launch_ms") or 0), r.get("frac") or 0, d["config"].get("setup_s") or 0))
PY
