#!/bin/bash
set -e
OUT=gpurun_out/r02_opt2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 tools/op_bench.py"
{
$B --hid 1 --level 1 --mode 2
$B --hid 1 --level 1 --mode 2 --opt bsr_nt=1
$B --hid 1 --level 1 --mode 2 --opt bsr_map=1
$B --hid 1 --level 1 --mode 2 --opt bsr_map=1 --opt bsr_nt=1
$B --hid 1 --level 1 --mode 2 --opt bsr_map=2 --opt bsr_sub=8
$B --hid 1 --level 1 --mode 2 --opt bsr_map=2 --opt bsr_sub=8 --opt bsr_nt=1
$B --hid 1 --level 1 --mode 2 --opt bsr_map=2 --opt bsr_sub=16 --opt bsr_nt=1
$B --hid 1 --level 1 --mode 2 --opt bsr_map=2 --opt bsr_sub=4 --opt bsr_nt=1
$B --hid 1 --level 0 --what cinv
$B --hid 1 --level 0 --what cinv --opt dense_map=1
$B --hid 1 --level 0 --what cinv --opt dense_map=2 --opt bsr_sub=2
} > $OUT/timings.jsonl 2> $OUT/timings.err
cat $OUT/timings.jsonl
for v in nt:"--opt bsr_nt=1" m1:"--opt bsr_map=1" m1nt:"--opt bsr_map=1 --opt bsr_nt=1" m2nt:"--opt bsr_map=2 --opt bsr_sub=8 --opt bsr_nt=1"; do
  tag=${v%%:*}; a=${v#*:}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f_$tag -- python3 tools/op_bench.py --hid 1 --level 1 --mode 2 $a --reps 10 > $OUT/pmc_f_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/r02_opt2"
for tag in ("nt", "m1", "m1nt", "m2nt"):
    acc = [0, 0.0]
    for f in glob.glob(os.path.join(out, "pmc_f_" + tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == "FETCH_SIZE" and "bsr_mfma<3" in r["Kernel_Name"]:
                acc[0] += 1; acc[1] += float(r["Counter_Value"])
    print(tag, "FETCH x2 = %.1f MB" % (2 * acc[1] / max(1, acc[0]) * 1024 / 1e6))
PY
