#!/bin/bash
# f32-preconditioner configuration sweep (one GPU call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_f32_sweep.txt
: > $OUT
run() {   # $1 cfg json, $2 extra bench args
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --engine-opts "f32_splitk=1" --cfg "$1" $2 > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1 $2" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-150s %-22s value=%7.0f iters=%s %s' % (sys.argv[1][60:], sys.argv[2], d['value'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" "$2" >> $OUT
}
B='"coarsening": [[4,8],[2,8],[2,8]], "smoother": "richardson", "setup": "device", "eo_levels": [0,1,2], "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1, "precond_precision": "f32"'
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}" "--streams 4"
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}" "--streams 5"
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}" "--nb 512 --streams 2"
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}" "--nb 512 --streams 3"
run "{$B, \"restart\": 4, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,8,0],[0,5,0],[0,14,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,8,0],[0,7,0],[0,14,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,7,0],[0,14,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,10,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0],[0,20,0]]}" ""
run "{$B, \"restart\": 3, \"cycle\": [[0,10,0],[0,8,0],[0,16,0]]}" ""
B3='"coarsening": [[4,8],[2,8]], "smoother": "richardson", "setup": "device", "eo_levels": [0,1], "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1, "precond_precision": "f32"'
run "{$B3, \"restart\": 3, \"cycle\": [[0,6,0],[0,5,0]]}" ""
run "{$B3, \"restart\": 3, \"cycle\": [[0,6,0],[0,3,0]]}" ""
cat $OUT
