#!/bin/bash
# round 3, sixth GPU session: time-skewed smoother order, relaxation sweeps, PMC passes
OUT=gpurun_out/${1:-r03f}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engine.py -m gpu -q -rP -p no:cacheprovider -k "time_skewed or config5 or reference_faithful or device_side_setup or synthetic or benchmarked" > $OUT/gputests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/gputests.log
grep -E "passed|failed|1024\^2:|reference-faithful cycle|FAILED|Error" $OUT/gputests.log | tail -30
if [ $rc -gt 1 ]; then exit $rc; fi
B="timeout -k 10 400 python bench.py --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
S="$B --workload synthetic --nb 64 --steps 3 --warmup 1"
$B --steps 8 --warmup 2 > $OUT/b_default.json 2> $OUT/b_default.err && \
$S --lattice 1024 > $OUT/b_synth1024.json 2> $OUT/b_synth1024.err && \
$S --lattice 1024 --engine-opts eo_skew=0 > $OUT/b_synth1024_noskew.json 2> $OUT/b_synth1024_noskew.err && \
$S --lattice 1024 --engine-opts eo_skew=128 > $OUT/b_synth1024_h128.json 2> $OUT/b_synth1024_h128.err && \
$S --lattice 1024 --engine-opts eo_skew=64,mfma3_tiles=1 > $OUT/b_synth1024_t1.json 2> $OUT/b_synth1024_t1.err && \
$S --lattice 512 > $OUT/b_synth512.json 2> $OUT/b_synth512.err && \
$S --lattice 512 --engine-opts eo_skew=0 > $OUT/b_synth512_noskew.json 2> $OUT/b_synth512_noskew.err
echo "bench rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    try:
        d = json.load(open(f))
        sb = d.get("step_breakdown_ms") or {}
        print("%-28s value %8.1f ms/step %7.2f iters %s launches %s dots %.2f axpy %.2f mvm %.2f coarsest %.2f P %.2f R %.2f other %.2f setup %.2f"
              % (f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["outer_iterations_max"],
                 sb.get("kernel_launches"), sb.get("dots", 0), sb.get("axpy", 0), sb.get("mvm", 0),
                 sb.get("coarsest", 0), sb.get("P", 0), sb.get("R", 0), sb.get("other", 0), d["config"]["setup_s"]))
        for r in d["kernel_rooflines"][:3]:
            print("      %-34s n=%4d avg %7.1f us total %7.2f ms frac %.3f" % (r["kernel"], r["launches_in_step"], r["avg_launch_ms"] * 1e3, r["step_ms"], r["frac"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python3 bench.py --steps 1 --warmup 0 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $P > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write rc=$?"
python3 tools/summarize_profile.py $OUT | tail -45
