#!/bin/bash
# round 3, third GPU session: asynchronous probe generation + direct level as the default; stream / width sweep
OUT=gpurun_out/${1:-r03c}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 600 python -m pytest tests/test_gpu_golden.py tests/test_probe_stream.py tests/test_gpu_two_ranks.py -m gpu -q -p no:cacheprovider > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/gputests.log
tail -n 15 $OUT/gputests.log
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil"
cfg() { python - "$@" <<PY
import json, sys
from deflatedmlmc_schwinger_amd import hierarchy
c = dict(hierarchy.TUNED_SOLVER_CFG_128)
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    c[k] = json.loads(v)
print(json.dumps(c))
PY
}
$B --streams 1 > $OUT/b_s1.json 2> $OUT/b_s1.err && \
$B --streams 2 > $OUT/b_s2.json 2> $OUT/b_s2.err && \
$B --streams 3 > $OUT/b_s3.json 2> $OUT/b_s3.err && \
$B --streams 2 --nb 128 > $OUT/b_s2_nb128.json 2> $OUT/b_s2_nb128.err && \
$B --streams 1 --nb 512 > $OUT/b_s1_nb512.json 2> $OUT/b_s1_nb512.err && \
$B --streams 1 --cfg "$(cfg 'cycle=[[0,7,0],[0,10,0]]')" > $OUT/b_s1_nu7.json 2> $OUT/b_s1_nu7.err && \
$B --streams 1 --cfg "$(cfg 'cycle=[[0,8,0],[0,10,0]]')" > $OUT/b_s1_nu8.json 2> $OUT/b_s1_nu8.err && \
$B --streams 1 --cfg "$(cfg 'cycle=[[0,10,0],[0,10,0]]')" > $OUT/b_s1_nu10.json 2> $OUT/b_s1_nu10.err && \
$B --streams 1 --cfg "$(cfg 'cycle=[[0,11,0],[0,10,0]]')" > $OUT/b_s1_nu11.json 2> $OUT/b_s1_nu11.err && \
$B --streams 1 --cfg "$(cfg 'restart=4')" > $OUT/b_s1_m4.json 2> $OUT/b_s1_m4.err && \
$B --streams 1 --cfg "$(cfg 'restart=2')" > $OUT/b_s1_m2.json 2> $OUT/b_s1_m2.err
echo "bench rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try:
        d = json.load(open(f))
        sb = d.get("step_breakdown_ms") or {}
        print("%-24s value %8.0f resident %8.0f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f P %.2f R %.2f other %.2f"
              % (f.split("/")[-1], d["value"], d["value_probes_resident"], d["ms_per_step"], d["config"]["outer_iterations_max"],
                 sb.get("kernel_launches"), sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0),
                 sb.get("coarsest", 0), sb.get("P", 0), sb.get("R", 0), sb.get("other", 0)))
    except Exception as e:
        print(f, "unreadable:", e)
PY
