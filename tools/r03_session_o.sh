#!/bin/bash
# round 3: K-cycle inner iteration in Gram form; whole suite
OUT=gpurun_out/${1:-r03o}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -12 $OUT/gputests.log
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
$B --workload synthetic --lattice 1024 --nb 64 --steps 3 --warmup 1 > $OUT/b_synth1024.json 2> $OUT/b_synth1024.err && \
$B --workload synthetic --lattice 512 --nb 64 --steps 3 --warmup 1 > $OUT/b_synth512.json 2> $OUT/b_synth512.err && \
$B --workload mlmc --steps 8 --warmup 2 > $OUT/b_mlmc.json 2> $OUT/b_mlmc.err && \
$B --steps 8 --warmup 2 > $OUT/b_default.json 2> $OUT/b_default.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    print("%-24s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f other %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0), sb.get("coarsest", 0), sb.get("other", 0)))
PY
