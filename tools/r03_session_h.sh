#!/bin/bash
# round 3, eighth GPU session: 16-stage dense variant; does any TCC read counter work under rocprofv3 here?
OUT=gpurun_out/${1:-r03h}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
$B > $OUT/b_g8.json 2> $OUT/b_g8.err && \
$B --engine-opts dense_stages=16 > $OUT/b_g16.json 2> $OUT/b_g16.err && \
$B --workload mlmc --streams 3 > $OUT/b_mlmc_s3.json 2> $OUT/b_mlmc_s3.err && \
$B --workload mlmc --streams 1 > $OUT/b_mlmc_s1.json 2> $OUT/b_mlmc_s1.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    r = [x for x in d["kernel_rooflines"] if "dense" in x["kernel"]][0]
    print("%-22s value %8.1f  dense avg %7.1f us  frac %.3f" % (f.split("/")[-1], d["value"], r["avg_launch_ms"] * 1e3, r["frac"]))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/tiny.py <<PY
import torch
x = torch.ones(1 << 26, device="cuda")
y = x * 2.0
torch.cuda.synchronize()
print(float(y[0]))
PY
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tiny -- python3 /tmp/tiny.py > $OUT/pmc_tiny.out 2> $OUT/pmc_tiny.err
echo "tiny FETCH_SIZE rc=$?"
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_tiny2 -- python3 /tmp/tiny.py > $OUT/pmc_tiny2.out 2> $OUT/pmc_tiny2.err
echo "tiny FETCH_SIZE (no kernel trace) rc=$?"
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tiny3 -- python3 /tmp/tiny.py > $OUT/pmc_tiny3.out 2> $OUT/pmc_tiny3.err
echo "tiny WRITE_SIZE rc=$?"
rocprofv3 --version 2>&1 | head -3
