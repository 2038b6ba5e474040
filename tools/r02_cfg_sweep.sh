#!/bin/bash
# solver-hierarchy shapes through bench.py (device-side setup makes each run cheap)
OUT=gpurun_out/r02_cfg_sweep.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" ${2:-} > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-150s %s value=%7.0f resident=%7.0f iters=%s setup=%.1fs %s' % (sys.argv[1], sys.argv[2] if len(sys.argv)>2 else '', d['value'], d['value_probes_resident'], c['outer_iterations_max'], c['solver']['setup_s'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" "${2:-}" >> $OUT
}
run '{"coarsening": [[4,8],[2,8]], "cycle": [[0,7,0],[0,7,0]], "restart": 6, "smoother": "richardson", "eig_tol": 1e-6}'
run '{"coarsening": [[4,8],[2,8]], "cycle": [[0,7,0],[0,7,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,12,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,16,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,9,0],[0,16,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,2],[0,12,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[2,8],[4,8]], "cycle": [[0,7,0],[0,7,0],[0,16,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[4,8]], "cycle": [[0,7,0],[0,14,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run '{"coarsening": [[4,8],[2,8]], "cycle": [[0,7,0],[0,7,0]], "restart": 6, "smoother": "richardson", "setup": "device"}' "--streams 3"
run '{"coarsening": [[4,8],[2,8]], "cycle": [[0,7,0],[0,7,0]], "restart": 6, "smoother": "richardson", "setup": "device"}' "--streams 1"
cat $OUT
