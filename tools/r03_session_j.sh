#!/bin/bash
# round 3: multi-rank rehearsal of bench.py on the one GPU (gloo transport and, with one rank, the nccl path
# incl. the engine's own RCCL all-reduce through the already-mapped library), then the whole GPU suite
OUT=gpurun_out/${1:-r03j}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="--steps 4 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
SW_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 $B > $OUT/b_2rank_gloo.json 2> $OUT/b_2rank_gloo.err
echo "2 ranks gloo rc=$?"
SW_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 $B > $OUT/b_4rank_gloo.json 2> $OUT/b_4rank_gloo.err
echo "4 ranks gloo rc=$?"
SW_FORCE_PROCESS_GROUP=1 SW_ENGINE_COMM=1 timeout -k 10 400 python bench.py --gpus 1 $B > $OUT/b_1rank_nccl_enginecomm.json 2> $OUT/b_1rank_nccl_enginecomm.err
echo "1 rank nccl + engine comm rc=$?"
grep -i "rccl\|nccl" $OUT/b_1rank_nccl_enginecomm.err | head -5
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try:
        d = json.load(open(f))
        print("%-36s n_gpus %d value %8.1f ms/step %7.2f iters %s trace %s" % (f.split("/")[-1], d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], d["config"]["trace_estimate"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
timeout -k 10 900 python -m pytest tests -m gpu -q -rP -p no:cacheprovider --durations=8 > $OUT/gputests.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/gputests.log
grep -E "passed|failed|FAILED|256 golden|strict mode|1024\^2:" $OUT/gputests.log | tail -12
