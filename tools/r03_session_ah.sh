#!/bin/bash
# round 3: kernel trace of the headline bench (every launch of the timed batches, in order)
OUT=gpurun_out/${1:-r03ah}
mkdir -p $OUT
export OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
ls $OUT/trace/*/
