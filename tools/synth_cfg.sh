#!/bin/bash
# Synthetic-lattice configuration A/B (one GPU call): SYN_L lattice extent, SYN_CFGS = ';'-separated
# "eo_levels|cycle" pairs, e.g. "[0]|[[0,6,0],[0,7,2],[0,7,2],[0,7,0],[0,16,0]]"; SYN_OPTS engine options.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=${SYN_L:-1024}
A0=${SYN_AGG0:-4}     # edge of the level-0 aggregates, in sites
depth="[[$A0,8]"; Lc=$((L/A0)); while [ $Lc -gt 16 ]; do depth="$depth,[2,8]"; Lc=$((Lc/2)); done; depth="$depth]"
IFS=';' read -ra CF <<< "$SYN_CFGS"
for c in "${CF[@]}"; do
  eo="${c%%|*}"; cyc="${c##*|}"
  cfg="{\"coarsening\": $depth, \"cycle\": $cyc, \"restart\": 3, \"eo_levels\": $eo, \"setup\": \"device\", \"setup_sweeps\": 3, \"setup_tol\": 0.1, \"setup_maxiter\": 32, \"setup_refine\": 1}"
  timeout -k 10 900 python3 bench.py --workload synthetic --lattice $L --nb ${SYN_NB:-64} --streams ${SYN_STREAMS:-1} --steps 3 --warmup 1 --no-large-stencil --cfg "$cfg" --engine-opts "${SYN_OPTS}" > gpurun_out/_s.json 2> gpurun_out/_s.err || { echo "FAILED $c"; tail -4 gpurun_out/_s.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/_s.json')); f=d.get('f32_preconditioner') or {}
print('L=%s eo=%s cyc=%s value=%.1f its=%s f32=%.1f/%s setup=%.1fs' % (sys.argv[1], sys.argv[2], sys.argv[3], d['value'], d['config']['outer_iterations_max'], f.get('value', 0.0), f.get('outer_iterations_max'), d['config']['solver']['setup_s']), {k: round(v,1) for k,v in d['step_breakdown_ms'].items()})" "$L" "$eo" "$cyc"
done
