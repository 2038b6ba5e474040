#!/bin/bash
# Batch width x concurrent streams on the default workload (one GPU call).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_batch_sweep.txt; : > $out
for cfg in "256 3" "256 4" "512 1" "512 2" "512 3" "768 1" "768 2" "1024 1" "1024 2"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --nb $1 --streams $2 --steps 4 --warmup 1 --no-cpu-baseline --no-large-stencil > gpurun_out/_b.json 2> gpurun_out/_b.err || { echo "nb=$1 streams=$2 FAILED" >> $out; tail -3 gpurun_out/_b.err >> $out; continue; }
  python3 - $1 $2 >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/_b.json"))
print("nb=%s streams=%s value=%.0f ms_per_step=%.2f its=%s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("outer_iterations_max")))
PY
done
cat $out
