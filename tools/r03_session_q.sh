#!/bin/bash
# round 3: small reference levels solved directly (MLMC coarse solves); whole suite, MLMC benches, drop-in flows
OUT=gpurun_out/${1:-r03q}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 $OUT/gputests.log
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
$B --workload mlmc --streams 1 > $OUT/b_mlmc_s1.json 2> $OUT/b_mlmc_s1.err && \
$B --workload mlmc --streams 2 > $OUT/b_mlmc_s2.json 2> $OUT/b_mlmc_s2.err && \
$B --workload mlmc --streams 3 > $OUT/b_mlmc_s3.json 2> $OUT/b_mlmc_s3.err && \
$B --workload mlmc --streams 1 --engine-opts direct_small=0 > $OUT/b_mlmc_s1_iter.json 2> $OUT/b_mlmc_s1_iter.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    sb = d.get("step_breakdown_ms") or {}
    print("%-24s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f other %.2f" % (
        f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"], sb.get("kernel_launches"),
        sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0), sb.get("coarsest", 0), sb.get("other", 0)))
PY
export SW_REPORT_PATH=$OUT/flows.jsonl
for e in 1 3; do
SW_ENGINES=$e python - <<PY
import contextlib, io, json, time
from deflatedmlmc_schwinger_amd import gateway
for name in ("G202", "G102"):
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        r = getattr(gateway, name)()
    print(name, "engines $e", "wall %.2f s" % (time.time() - t0), "trace", complex(r["trace"]))
PY
done
python - <<PY
import json
for l in open("$OUT/flows.jsonl"):
    r = json.loads(l)
    print(r["kind"], "elapsed %.2f" % r["elapsed_s"], "loop solved/s %.0f" % r.get("probe_loop_solved_per_s", 0), "probes", r.get("probes_solved"))
PY
