#!/bin/bash
OUT=gpurun_out/r02_opt5
mkdir -p $OUT
{
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30 --opt dense_stages=4
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30 --opt dense_stages=8
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30 --opt dense_stages=8 --opt mfma_tiles=2
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30 --opt dense_stages=4 --opt mfma_tiles=2
python3 tools/op_bench.py --hid 1 --level 1 --reps 100 --mode 2 --opt bsr_stages=4 --opt mfma_tiles=2
python3 tools/op_bench.py --hid 1 --level 1 --reps 100 --mode 2 --opt bsr_stages=4
} > $OUT/timings2.jsonl 2> $OUT/timings.err
cat $OUT/timings2.jsonl
