cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/s2y
timeout -k 10 120 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -s -k "block_level_solved_exactly" > gpurun_out/s2y/t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s2y/t.log
grep -n "iterations direct\|passed\|failed\|Error\|assert" gpurun_out/s2y/t.log | head -12
CFG='{"coarsening": [[8,8],[2,8]], "cycle": [[0,9,0],[0,10,0]], "smoother": "richardson", "eo_levels": [0,1], "restart": 3, "setup": "device", "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1, "direct_levels": [1]}'
timeout -k 10 150 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-large-stencil --cfg "$CFG" > gpurun_out/s2y/bench_direct.json 2> gpurun_out/s2y/bench_direct.err || tail -5 gpurun_out/s2y/bench_direct.err
python3 -c "
import json; d=json.load(open('gpurun_out/s2y/bench_direct.json')); f=d['f32_preconditioner']
print('direct level 1:', round(d['value']), d['config']['outer_iterations_max'], 'f32', round(f['value']), f['outer_iterations_max'], 'setup_s', round(d['config']['setup_s'],1))
print({k: round(v,2) for k,v in d['step_breakdown_ms'].items()})
print([(k['kernel'], round(k['avg_launch_ms']*1e3,1), k['launches_in_step'], round(k['step_ms'],2)) for k in d['kernel_rooflines']])"
