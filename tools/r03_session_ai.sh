#!/bin/bash
# round 3, final state: batch width x streams once more (product form, Gram cycles, direct block level)
OUT=gpurun_out/${1:-r03ai}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
$B --nb 256 --streams 1 > $OUT/b_256x1.json 2> $OUT/b_256x1.err
$B --nb 512 --streams 1 > $OUT/b_512x1.json 2> $OUT/b_512x1.err
$B --nb 128 --streams 2 > $OUT/b_128x2.json 2> $OUT/b_128x2.err
$B --nb 256 --streams 2 > $OUT/b_256x2.json 2> $OUT/b_256x2.err
$B --nb 128 --streams 1 > $OUT/b_128x1.json 2> $OUT/b_128x1.err
$B --nb 384 --streams 1 > $OUT/b_384x1.json 2> $OUT/b_384x1.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    print("%-16s value %8.1f ms/step %7.2f iters %s launches %s mvm %.2f coarsest %.2f dots %.2f axpy %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("mvm", 0), sb.get("coarsest", 0), sb.get("dots", 0), sb.get("axpy", 0)))
PY
