#!/bin/bash
# round 3: x-tile width of the lattice walk on the synthetic 1024^2 lattice (cold launches run at 2-3 TB/s)
OUT=gpurun_out/${1:-r03aq}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --workload synthetic --lattice 1024 --nb 128 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for t in 256 64 128 512 1024; do
  $B --engine-opts "stencil_tile=$t" > $OUT/b_tile$t.json 2> $OUT/b_tile$t.err || { echo "tile $t failed"; tail -3 $OUT/b_tile$t.err; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try: d = json.load(open(f))
    except Exception: print(f, "unreadable"); continue
    sb = d.get("step_breakdown_ms") or {}
    ks = {r["kernel"]: r for r in d.get("kernel_rooflines", [])}
    print("%-16s value %8.1f ms/step %7.2f iters %s mvm %.2f | strips %.1f us  S-op %.1f us" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("mvm", 0),
        1e3 * ks["k_schur_step"]["avg_launch_ms"], 1e3 * ks["k_schur_step<0/1> (S x, b' - S x)"]["avg_launch_ms"]))
PY
