#!/bin/bash
OUT=gpurun_out/r02_cfg_sweep8b.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-170s value=%7.0f iters=%s %s' % (sys.argv[1], d['value'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" >> $OUT
}
B='"coarsening": [[4,8],[2,8],[2,8]], "smoother": "richardson", "setup": "device", "eo_levels": [0,1,2], "cycle": [[0,6,0],[0,5,0],[0,14,0]]'
run "{$B, \"restart\": 1}"
run "{$B, \"restart\": 2}"
run "{$B, \"restart\": 3}"
run "{$B, \"restart\": 4}"
cat $OUT
