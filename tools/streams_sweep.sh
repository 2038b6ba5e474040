#!/bin/bash
# bench value against concurrent probe batches per GPU and batch width (all-fp64 default workload)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in ${SWEEP:-"256 2" "256 3" "256 4" "256 6" "128 4" "128 6" "128 8" "512 2"}; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --no-f32-line --nb $1 --streams $2 > gpurun_out/_b.json 2> gpurun_out/_b.err || { echo "nb=$1 streams=$2 FAILED"; tail -3 gpurun_out/_b.err; continue; }
  python3 - $1 $2 <<'PY'
import json, sys
d = json.load(open("gpurun_out/_b.json"))
print("nb=%s streams=%s value=%.0f ms_per_step=%.2f its=%s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("outer_iterations_max")), flush=True)
PY
done
