#!/bin/bash
# A/B of engine options on the default bench: AB_OPTS="a=1 b=2,c=3" (space-separated option sets)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in "" $AB_OPTS; do
  timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --no-f32-line ${AB_ARGS} --engine-opts "$o" > gpurun_out/_b.json 2> gpurun_out/_b.err || { echo "opts=$o FAILED"; tail -5 gpurun_out/_b.err; continue; }
  python3 - "$o" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_b.json"))
print("opts=%s value=%.0f its=%s" % (sys.argv[1], d["value"], d["config"].get("outer_iterations_max")))
print("   ", {k: round(v, 2) for k, v in d["step_breakdown_ms"].items()})
print("   ", "  ".join("%s %.1fus" % (k["kernel"].replace("k_bsr_mfma", "bsr"), k["avg_launch_ms"] * 1e3) for k in d["kernel_rooflines"]))
PY
done
