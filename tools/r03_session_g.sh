#!/bin/bash
# round 3, seventh GPU session: dense Schur-inverse kernel variants (block order x tiles x pipeline depth); PMC retry
OUT=gpurun_out/${1:-r03g}
mkdir -p $OUT
export OMP_NUM_THREADS=1
B="timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs"
for m in 0 1 2; do for t in 1 2; do for g in 4 8; do
  $B --engine-opts dense_map=$m,mfma3_tiles=$t,dense_stages=$g > $OUT/b_m${m}_t${t}_g${g}.json 2> $OUT/b_m${m}_t${t}_g${g}.err || exit 1
done; done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d = json.load(open(f))
    r = [x for x in d["kernel_rooflines"] if "dense" in x["kernel"]][0]
    print("%-22s value %8.1f  dense avg %7.1f us" % (f.split("/")[-1], d["value"], r["avg_launch_ms"] * 1e3))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python3 bench.py --steps 1 --warmup 0 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_rdreq -- $P > $OUT/bench_pmc_rdreq.json 2> $OUT/pmc_rdreq.err
echo "pmc rdreq rc=$?"
ls $OUT/pmc_rdreq/*/ 2>/dev/null | head
