#!/bin/bash
# Run on the GPU box: bench line, rocprofv3 kernel stats, and separate PMC passes (HBM traffic).
set -e
R=${1:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 6 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
# per-kernel durations (one batch at a time per GPU is the default since round 3), no 1024^2 stencil
# point (it would mix into k_stencil<0>), none of the secondary configurations
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 4 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
python3 tools/summarize_profile.py $OUT
