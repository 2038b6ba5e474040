#!/bin/bash
OUT=gpurun_out/r02_cfg_sweep2.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" --engine-opts "${2:-}" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-140s %-22s value=%7.0f resident=%7.0f iters=%s %s' % (sys.argv[1], sys.argv[2], d['value'], d['value_probes_resident'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" "${2:-}" >> $OUT
}
C4='{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,16,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run "$C4" "mfma_small_tiles=0"
run "$C4" "mfma_small_tiles=2"
run "$C4" "mfma_small_tiles=1"
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,20,0]], "restart": 6, "smoother": "richardson", "setup": "device"}' "mfma_small_tiles=2"
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,6,0],[0,20,0]], "restart": 6, "smoother": "richardson", "setup": "device"}' "mfma_small_tiles=2"
run '{"coarsening": [[4,8],[2,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,16,0],[0,16,0]], "restart": 6, "smoother": "richardson", "setup": "device"}' "mfma_small_tiles=2"
run '{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,16,0]], "restart": 6, "smoother": "richardson", "eig_tol": 1e-6}' "mfma_small_tiles=2"
cat $OUT
