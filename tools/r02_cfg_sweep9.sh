#!/bin/bash
OUT=gpurun_out/r02_cfg_sweep9.txt
: > $OUT
run() {
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$1" > gpurun_out/r02_cfg_tmp.json 2> gpurun_out/r02_cfg_tmp.err || { echo "FAILED $1" >> $OUT; tail -3 gpurun_out/r02_cfg_tmp.err >> $OUT; return; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/r02_cfg_tmp.json'));c=d['config']
print('%-170s value=%7.0f iters=%s %s' % (sys.argv[1], d['value'], c['outer_iterations_max'], {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$1" >> $OUT
}
B='"coarsening": [[4,8],[2,8],[2,8]], "smoother": "richardson", "setup": "device", "eo_levels": [0,1,2], "restart": 3'
run "{$B, \"cycle\": [[0,6,0],[0,5,0],[0,14,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,2],[0,14,0]]}"
run "{$B, \"cycle\": [[0,6,0],[0,5,2],[0,10,0]]}"
run "{$B, \"cycle\": [[0,5,0],[0,4,2],[0,10,0]]}"
cat $OUT
for L in 512 1024; do timeout -k 10 400 python3 bench.py --workload synthetic --lattice $L --nb 64 --streams 1 --steps 3 --warmup 1 > gpurun_out/r02_synth$L.json 2> gpurun_out/r02_synth$L.err || tail -3 gpurun_out/r02_synth$L.err; done
timeout -k 10 400 python3 bench.py --workload synthetic --lattice 1024 --nb 64 --streams 2 --steps 3 --warmup 1 > gpurun_out/r02_synth1024_s2.json 2> gpurun_out/r02_synth1024_s2.err
python3 -c "
import json
for f in ('r02_synth512','r02_synth1024','r02_synth1024_s2'):
    d=json.load(open('gpurun_out/%s.json'%f));print(f, round(d['value'],1), d['config']['outer_iterations_max'], round(d['config']['solver']['setup_s'],1), d['roofline']['kernel'], round(d['roofline']['frac'],3))"
