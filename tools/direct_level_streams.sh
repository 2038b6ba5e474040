#!/bin/bash
# direct (dense Schur inverse) solve of block level 1 against concurrent batches per GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG='{"coarsening": [[8,8],[2,8]], "cycle": [[0,9,0],[0,10,0]], "smoother": "richardson", "eo_levels": [0,1], "restart": 3, "setup": "device", "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1, "direct_levels": [1]}'
for st in 1 2 4; do
  for opt in "eo_direct=1" "eo_direct=0"; do
    timeout -k 10 100 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --no-f32-line --streams $st --cfg "$CFG" --engine-opts "$opt" > gpurun_out/_b.json 2> gpurun_out/_b.err || { echo "streams=$st $opt FAILED"; continue; }
    python3 -c "
import json; d=json.load(open('gpurun_out/_b.json')); print('streams=$st $opt', round(d['value']), d['config']['outer_iterations_max'])"
  done
done
