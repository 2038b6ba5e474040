#!/bin/bash
OUT=gpurun_out/r02_opt4
mkdir -p $OUT
B="python3 tools/op_bench.py --hid 1 --level 1 --mode 2 --reps 100"
{
$B
$B --opt bsr_stagger=1 --opt bsr_stagger_mode=1
$B --opt bsr_stagger=2 --opt bsr_stagger_mode=1
$B --opt bsr_stagger=3 --opt bsr_stagger_mode=1
$B --opt bsr_stagger=1 --opt bsr_stagger_mode=2
$B --opt bsr_stagger=2 --opt bsr_stagger_mode=2
$B --opt bsr_stagger=3 --opt bsr_stagger_mode=2
$B --opt bsr_map=0 --opt bsr_stagger=2 --opt bsr_stagger_mode=1
$B --opt bsr_map=0 --opt bsr_stagger=2 --opt bsr_stagger_mode=2
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30 --opt dense_stagger=8 --opt bsr_stagger_mode=1
python3 tools/op_bench.py --hid 1 --level 0 --what cinv --reps 30 --opt dense_stagger=8 --opt bsr_stagger_mode=2
} > $OUT/timings.jsonl 2> $OUT/timings.err
cat $OUT/timings.jsonl
