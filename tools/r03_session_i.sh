#!/bin/bash
# PMC passes (FETCH_SIZE / WRITE_SIZE) of the default workload
OUT=gpurun_out/${1:-r03i}
mkdir -p $OUT
export OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python3 bench.py --steps 1 --warmup 0 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $P > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $P > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write rc=$?"
python3 tools/summarize_profile.py $OUT | tail -60
