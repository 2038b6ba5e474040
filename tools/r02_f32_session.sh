#!/bin/bash
# f32 preconditioner: parity test + A/B bench (one GPU call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "${F32_TESTS:-single_precision or even_odd}" > gpurun_out/f32_test.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/f32_test.log
for o in ${F32_OPTS:-"" "precond_f32=1"}; do
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-large-stencil --engine-opts "$o" > gpurun_out/_b.json 2> gpurun_out/_b.err || { echo "opts=$o FAILED"; tail -5 gpurun_out/_b.err; continue; }
  python3 - "$o" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_b.json"))
print("opts=%s value=%.0f its=%s" % (sys.argv[1], d["value"], d["config"].get("outer_iterations_max")))
print("   ", {k: round(v, 2) for k, v in d["step_breakdown_ms"].items()})
for k in d["kernel_rooflines"]:
    print("   ", k["kernel"], round(k["avg_launch_ms"] * 1e3, 1), "us", round(k["frac"], 3))
PY
done
