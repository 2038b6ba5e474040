#!/bin/bash
OUT=gpurun_out/r02_cfg_sweep7.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" ${2:-} > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-160s %s value=%7.0f iters=%s %s' % (sys.argv[1], sys.argv[2] if len(sys.argv)>2 else '', d['value'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" "${2:-}" >> $OUT
}
B='"coarsening": [[4,8],[2,8],[2,8]], "smoother": "richardson", "restart": 6, "setup": "device", "eo_levels": [0,1,2]'
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,12,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,6,0],[0,12,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,16,0]]}"
run "{$B, \"cycle\": [[0,5,0],[0,5,0],[0,12,0]]}"
run "{$B, \"cycle\": [[0,7,0],[0,5,0],[0,12,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,4,0],[0,12,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,12,0]], \"restart\": 5}"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,12,0]]}" "--streams 4"
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,12,0]]}" "--streams 2"
run '{"coarsening": [[4,8],[2,8]], "smoother": "richardson", "restart": 6, "setup": "device", "eo_levels": [0,1], "cycle": [[0,6,0],[0,5,0]]}'
cat $OUT
