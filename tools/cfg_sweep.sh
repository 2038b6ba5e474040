#!/bin/bash
# Solver-configuration sweep through bench.py --cfg (one GPU call).
#   SWEEP_COARSENING  JSON list, default the tuned [[8,8],[2,8]]
#   SWEEP_CYCLES      space-separated Schur-step lists, one number per smoothed level: "12,10 10,10"
#   SWEEP_EXTRA       extra JSON members (e.g. '"precond_precision": "f32"')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/cfg_sweep.txt
: > $OUT
CO=${SWEEP_COARSENING:-[[8,8],[2,8]]}
for c in ${SWEEP_CYCLES:-12,10}; do
  cyc=$(python3 -c "import sys; print([[0,int(v),0] for v in sys.argv[1].split(',')])" "$c")
  eo=$(python3 -c "import sys; print(list(range(len(sys.argv[1].split(',')))))" "$c")
  cfg="{\"coarsening\": $CO, \"cycle\": $cyc, \"eo_levels\": $eo, \"smoother\": \"richardson\", \"setup\": \"device\", \"setup_sweeps\": 3, \"setup_tol\": 0.1, \"setup_maxiter\": 32, \"setup_refine\": 1, \"restart\": 3${SWEEP_EXTRA:+, $SWEEP_EXTRA}}"
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$cfg" > gpurun_out/_c.json 2> gpurun_out/_c.err || { echo "FAILED $c" >> $OUT; tail -3 gpurun_out/_c.err >> $OUT; continue; }
  python3 -c "
import json,sys;d=json.load(open('gpurun_out/_c.json'));c=d['config'];f=d.get('f32_preconditioner') or {}
print('%s steps %-10s value=%7.0f iters=%s f32=%.0f/%s %s' % (sys.argv[2], sys.argv[1], d['value'], c['outer_iterations_max'], f.get('value', 0), f.get('outer_iterations_max'), {k:round(v,2) for k,v in d['step_breakdown_ms'].items()}))" "$c" "$CO" >> $OUT
done
cat $OUT
