#!/bin/bash
OUT=gpurun_out/r02_synth_sweep3.txt
: > $OUT
run() {
  timeout -k 10 400 python3 bench.py --workload synthetic --lattice ${2:-1024} --nb 64 --streams ${3:-1} --steps 3 --warmup 1 --cfg "$1" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-120s L=%s streams=%s value=%6.1f iters=%s setup=%.1f %s' % (sys.argv[1][170:], sys.argv[2], sys.argv[3], d['value'], c['outer_iterations_max'], c['solver']['setup_s'], {k:round(v,1) for k,v in d['step_breakdown_ms'].items()}))" "$1" "${2:-1024}" "${3:-1}" >> $OUT
}
S='"smoother": "richardson", "setup": "device", "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "eo_levels": [0], "restart": 3,                     '
C5='"coarsening": [[4,8],[2,8],[2,8],[2,8],[2,8]]'
run "{$S $C5, \"cycle\": [[0,6,0],[0,7,2],[0,7,0],[0,7,0],[0,16,0]]}"
run "{$S $C5, \"cycle\": [[0,6,0],[0,7,2],[0,7,2],[0,7,0],[0,16,0]]}"
run "{$S $C5, \"cycle\": [[0,6,0],[0,7,3],[0,7,0],[0,7,0],[0,16,0]]}"
run "{$S $C5, \"cycle\": [[0,6,2],[0,7,0],[0,7,0],[0,7,0],[0,16,0]]}"
run "{$S $C5, \"cycle\": [[0,6,0],[0,7,0],[0,7,2],[0,7,0],[0,16,0]]}"
run "{$S $C5, \"cycle\": [[0,6,0],[0,5,2],[0,7,0],[0,7,0],[0,16,0]]}"
run "{$S \"coarsening\": [[4,8],[2,8],[2,8],[2,8],[2,8],[2,8]], \"cycle\": [[0,6,0],[0,7,2],[0,7,0],[0,7,0],[0,7,0],[0,16,0]]}"
run "{$S $C5, \"cycle\": [[0,6,0],[0,7,2],[0,7,0],[0,7,0],[0,16,0]]}" 1024 2
run "{$S \"coarsening\": [[4,8],[2,8],[2,8],[2,8]], \"cycle\": [[0,6,0],[0,7,2],[0,7,0],[0,16,0]]}" 512
cat $OUT
