#!/bin/bash
OUT=gpurun_out/r02_cfg_sweep3.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" --engine-opts "${2:-}" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-140s %-22s value=%7.0f resident=%7.0f iters=%s %s' % (sys.argv[1], sys.argv[2], d['value'], d['value_probes_resident'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" "${2:-}" >> $OUT
}
C3='{"coarsening": [[4,8],[2,8]], "cycle": [[0,7,0],[0,7,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
C4='{"coarsening": [[4,8],[2,8],[2,8]], "cycle": [[0,7,0],[0,7,0],[0,16,0]], "restart": 6, "smoother": "richardson", "setup": "device"}'
run "$C3" ""
run "$C4" ""
run "$C3" "dense_stages=2,bsr_stages=2"
run "$C3" ""
run "$C4" ""
cat $OUT
