#!/bin/bash
OUT=gpurun_out/r02_opt6.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-large-stencil --engine-opts "$1" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-30s value=%7.0f iters=%s %s %s' % (sys.argv[1], d['value'], c['outer_iterations_max'], [(k['kernel'][:24], round(k['avg_launch_ms']*1e3,1)) for k in d['kernel_rooflines'][:5]], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" >> $OUT
}
run "bsr_splitk=0"
run "bsr_splitk=1"
run "bsr_splitk=0"
run "bsr_splitk=1"
cat $OUT
