#!/bin/bash
# 2-rank rehearsal of bench.py on ONE GPU (gloo transport): exercises the sharded probe stream,
# the per-step all-reduce and the max-over-ranks timing.  Not a scaling measurement.
export SW_DIST_BACKEND=gloo
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline
