#!/bin/bash
# wide probe batches with the even-odd smoother of the stencil level run on groups of 64-probe chunks (eo_chunk)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# WIDE_CFGS: configurations "nb streams engine-opts", separated by ";"
IFS=";" read -ra CFGS <<< "${WIDE_CFGS:-768 1 eo_chunk=4}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-large-stencil --no-f32-line --nb $1 --streams $2 --engine-opts "$3" > gpurun_out/_b.json 2> gpurun_out/_b.err || { echo "$cfg FAILED"; tail -3 gpurun_out/_b.err; continue; }
  python3 - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_b.json"))
print("%-22s value=%.0f ms_per_step=%.2f its=%s %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("outer_iterations_max"), {k: round(v, 2) for k, v in d["step_breakdown_ms"].items()}), flush=True)
PY
done
