#!/bin/bash
# round 3: dense Schur inverse streamed with non-temporal operator loads (dense_nt): does keeping the 67 MB
# (the dense_nt option was removed after this session: profiles/r03_ab_sessions.txt, r03ak)
# matrix out of the Infinity Cache help the vectors of the iteration?
OUT=gpurun_out/${1:-r03ak}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
i=0
for v in 1 0 1 0; do
  i=$((i+1))
  $B --engine-opts "dense_nt=$v" > $OUT/b_nt${v}_$i.json 2> $OUT/b_nt${v}_$i.err || { tail -5 $OUT/b_nt${v}_$i.err; exit 1; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    ks = {r["kernel"]: r for r in d.get("kernel_rooflines", [])}
    print("%-16s value %8.1f ms/step %7.2f mvm %.2f coarsest %.2f | schur %.1f us  dense %.1f us  S-op %.1f us" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], sb.get("mvm", 0), sb.get("coarsest", 0),
        1e3 * ks["k_schur_step"]["avg_launch_ms"], 1e3 * ks["k_bsr_mfma(dense coarsest)"]["avg_launch_ms"],
        1e3 * ks["k_schur_step<0/1> (S x, b' - S x)"]["avg_launch_ms"]))
PY
