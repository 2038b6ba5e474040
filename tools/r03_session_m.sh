#!/bin/bash
# round 3: restart cycles in Gram-matrix form
OUT=gpurun_out/${1:-r03m}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_engine.py -m gpu -q -p no:cacheprovider -x -k "gram or benchmarked or golden_128 or reduced or stale or nonconvergence or zero_rhs or solve_reaches or strict or hutchinson_probes" > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -25 $OUT/gputests.log
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
$B > $OUT/b_gram.json 2> $OUT/b_gram.err && \
$B --engine-opts gram_cycle=0 > $OUT/b_arnoldi.json 2> $OUT/b_arnoldi.err && \
$B --engine-opts stop_factor=0.1 > $OUT/b_gram_strict.json 2> $OUT/b_gram_strict.err && \
$B --workload synthetic --lattice 1024 --nb 64 --steps 3 --warmup 1 > $OUT/b_synth1024_gram.json 2> $OUT/b_synth1024_gram.err && \
$B --workload synthetic --lattice 1024 --nb 64 --steps 3 --warmup 1 --engine-opts gram_cycle=0 > $OUT/b_synth1024_arnoldi.json 2> $OUT/b_synth1024_arnoldi.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    print("%-30s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f other %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0), sb.get("coarsest", 0), sb.get("other", 0)))
PY
