#!/bin/bash
# round 3, fourth GPU session: three-product MFMA kernel
OUT=gpurun_out/${1:-r03d}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_golden.py -m gpu -q -p no:cacheprovider -k "three_product or golden or benchmarked or config2 or dense_schur" > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/gputests.log
tail -n 25 $OUT/gputests.log
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil"
$B > $OUT/b_3m.json 2> $OUT/b_3m.err && \
$B --engine-opts mfma_3m=0 > $OUT/b_4m.json 2> $OUT/b_4m.err && \
$B --engine-opts mfma3_tiles=1 > $OUT/b_3m_t1.json 2> $OUT/b_3m_t1.err && \
$B --engine-opts mfma3_tiles=2 > $OUT/b_3m_t2.json 2> $OUT/b_3m_t2.err && \
$B --engine-opts mfma3_tiles=4 > $OUT/b_3m_t4.json 2> $OUT/b_3m_t4.err && \
$B --engine-opts dense_stages=4 > $OUT/b_3m_ds4.json 2> $OUT/b_3m_ds4.err && \
$B --workload mlmc > $OUT/b_mlmc.json 2> $OUT/b_mlmc.err && \
$B --workload config2 --steps 2 --warmup 1 > $OUT/b_config2.json 2> $OUT/b_config2.err && \
$B --workload config2 --steps 2 --warmup 1 --engine-opts mfma_3m=0 > $OUT/b_config2_4m.json 2> $OUT/b_config2_4m.err
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
        for r in d["kernel_rooflines"]:
            if "mfma" in r["kernel"]:
                print("      %-34s n=%4d avg %6.1f us frac %.3f" % (r["kernel"], r["launches_in_step"], r["avg_launch_ms"] * 1e3, r["frac"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
