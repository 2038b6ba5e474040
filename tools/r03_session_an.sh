#!/bin/bash
# round 3: batch widths around the Infinity Cache capacity (the working set of an iteration is ~290 MB at 256 probes)
OUT=gpurun_out/${1:-r03an}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for nb in 256 192 320 192 256 128; do
  $B --nb $nb > $OUT/b_${nb}_$RANDOM.json 2> $OUT/b_${nb}.err || { tail -3 $OUT/b_${nb}.err; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    ks = {r["kernel"]: r for r in d.get("kernel_rooflines", [])}
    print("%-20s value %8.1f ms/step %7.2f iters %s launches %s | schur %.1f us dense %.1f us S-op %.1f us" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        1e3 * ks["k_schur_step"]["avg_launch_ms"], 1e3 * ks["k_bsr_mfma(dense coarsest)"]["avg_launch_ms"],
        1e3 * ks["k_schur_step<0/1> (S x, b' - S x)"]["avg_launch_ms"]))
PY
