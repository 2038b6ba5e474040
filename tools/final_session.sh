#!/bin/bash
# Round-2 evidence in one GPU call: profile of the default bench, the other workloads, synthetic lattices.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02 > gpurun_out/r02_profile.log 2>&1
timeout -k 10 400 python3 bench.py --workload config2 --steps 4 --warmup 1 --no-large-stencil > gpurun_out/r02_bench_config2.json 2> gpurun_out/r02_bench_config2.err
timeout -k 10 300 python3 bench.py --workload mlmc --steps 6 --warmup 2 --no-large-stencil > gpurun_out/r02_bench_mlmc.json 2> gpurun_out/r02_bench_mlmc.err
for L in 512 1024; do
  timeout -k 10 400 python3 bench.py --workload synthetic --lattice $L --nb 64 --streams 1 --steps 3 --warmup 1 > gpurun_out/r02_synth$L.json 2> gpurun_out/r02_synth$L.err
done
timeout -k 10 400 python3 bench.py --workload synthetic --lattice 1024 --nb 64 --streams 2 --steps 3 --warmup 1 --no-large-stencil > gpurun_out/r02_synth1024_s2.json 2> gpurun_out/r02_synth1024_s2.err
python3 - <<'PY'
import json
for f in ("r02/bench", "r02_bench_config2", "r02_bench_mlmc", "r02_synth512", "r02_synth1024", "r02_synth1024_s2"):
    d = json.load(open("gpurun_out/%s.json" % f))
    f32 = d.get("f32_preconditioner") or {}
    print(f, round(d["value"], 1), d["config"]["outer_iterations_max"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
          "f32:", round(f32.get("value", 0.0), 1), f32.get("outer_iterations_max"))
PY
# Schur steps on the lattice level of the 1024^2 hierarchy (default 14) with the even-odd reduced outer solve
for nu in 10 12; do
  cfg=$(python3 - $nu <<'PY'
import json, sys
nu = int(sys.argv[1])
depth = [[8, 8]] + [[2, 8]] * 3
cyc = [[0, nu, 0], [0, 10, 2], [0, 8, 0], [0, 14, 0]]
print(json.dumps({"coarsening": depth, "cycle": cyc, "restart": 3, "eo_levels": [0, 1, 2, 3], "setup": "device",
                  "setup_sweeps": 3, "setup_tol": 0.1, "setup_maxiter": 32, "setup_refine": 1}))
PY
)
  timeout -k 10 300 python3 bench.py --workload synthetic --lattice 1024 --nb 64 --streams 1 --steps 3 --warmup 1 --no-large-stencil --cfg "$cfg" > gpurun_out/r02_synth1024_nu$nu.json 2> gpurun_out/r02_synth1024_nu$nu.err || echo "nu=$nu failed"
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/r02_synth1024_nu$nu.json')); f=d.get('f32_preconditioner') or {}
print('1024^2 nu0=$nu', round(d['value'],1), d['config']['outer_iterations_max'], 'f32', round(f.get('value',0),1), f.get('outer_iterations_max'))"
done
