#!/bin/bash
# Solver-configuration sweep through bench.py --cfg (one GPU call).  SWEEP_CYCLES: space-separated
# cycle lists like "6,5,14 8,5,14"; SWEEP_EXTRA: extra JSON members (e.g. '"precond_precision": "f32"').
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/cfg_sweep.txt
: > $OUT
B='"coarsening": [[4,8],[2,8],[2,8]], "smoother": "richardson", "setup": "device", "eo_levels": [0,1,2], "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1, "restart": 3'
for c in ${SWEEP_CYCLES:-6,5,14}; do
  IFS=, read a b d <<< "$c"
  cfg="{$B${SWEEP_EXTRA:+, $SWEEP_EXTRA}, \"cycle\": [[0,$a,0],[0,$b,0],[0,$d,0]]}"
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --no-f32-line --cfg "$cfg" > gpurun_out/_c.json 2> gpurun_out/_c.err || { echo "FAILED $c" >> $OUT; tail -3 gpurun_out/_c.err >> $OUT; continue; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/_c.json'));c=d['config']
print('cycle %-10s %s value=%7.0f iters=%s %s' % (sys.argv[1], sys.argv[2], d['value'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$c" "${SWEEP_EXTRA}" >> $OUT
done
cat $OUT
