#!/bin/bash
OUT=gpurun_out/r02_cfg_sweep6.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-175s value=%7.0f iters=%s %s' % (sys.argv[1], d['value'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" >> $OUT
}
B='"coarsening": [[4,8],[2,8],[2,8]], "smoother": "richardson", "restart": 6, "setup": "device"'
run "{$B, \"cycle\": [[0,6,0],[0,7,0],[0,16,0]], \"eo_levels\": [0]}"
run "{$B, \"cycle\": [[0,6,0],[0,4,0],[0,16,0]], \"eo_levels\": [0,1]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,16,0]], \"eo_levels\": [0,1]}"
run "{$B, \"cycle\": [[0,6,0],[0,6,0],[0,16,0]], \"eo_levels\": [0,1]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,8,0]], \"eo_levels\": [0,1,2]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,10,0]], \"eo_levels\": [0,1,2]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,12,0]], \"eo_levels\": [0,1,2]}"
cat $OUT
