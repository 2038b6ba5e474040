#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "single_precision" > gpurun_out/f32_test.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/f32_test.log
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-large-stencil --no-cpu-baseline > gpurun_out/f32_default.json 2> gpurun_out/f32_default.err || tail -5 gpurun_out/f32_default.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/f32_default.json"))
print("default value", round(d["value"]), "its", d["config"]["outer_iterations_max"], "f32:", d["f32_preconditioner"])
PY
for o in "" "precond_f32=1"; do
for L in 512 1024; do
  timeout -k 10 400 python3 bench.py --workload synthetic --lattice $L --nb 64 --streams 1 --steps 3 --warmup 1 --no-f32-line --engine-opts "$o" > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "L=$L opts=$o FAILED"; tail -5 gpurun_out/_s.err; continue; }
  python3 - "$L" "$o" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_s.json"))
print("L=%s opts=%s value=%.1f its=%s" % (sys.argv[1], sys.argv[2], d["value"], d["config"].get("outer_iterations_max")), {k: round(v, 2) for k, v in d["step_breakdown_ms"].items()})
PY
done
done
