#!/bin/bash
# Three rocprofv3 runs of one bench.py command on the GPU box: kernel trace (+ stats), --pmc FETCH_SIZE, --pmc WRITE_SIZE
# (counters in passes of their own, the program directly after `--`), then tools/summarize_pmc.py.
#   usage: tools/profile_pmc.sh OUTDIR bench.py-arguments...
set -e
OUT=$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
COMMON="--no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" $COMMON > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py "$@" $COMMON > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py "$@" $COMMON > $OUT/bench_write.json 2> $OUT/pmc_write.err
python3 tools/summarize_pmc.py $OUT
# the raw traces are large: keep the tables only
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
