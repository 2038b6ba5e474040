#!/bin/bash
set -e
OUT=gpurun_out/r02_opt3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/op_bench.py --what cinv --hid 1 --level 0 > $OUT/warm.log 2>&1
for v in d0:"--what cinv --opt dense_map=0" d3:"--what cinv --opt dense_map=3" d1:"--what cinv --opt dense_map=1" d2:"--what cinv --opt dense_map=2 --opt bsr_sub=2" l0:"--level 1 --mode 2 --opt bsr_map=0" l1:"--level 1 --mode 2 --opt bsr_map=1" l3:"--level 1 --mode 2 --opt bsr_map=3"; do
  tag=${v%%:*}; a=${v#*:}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f_$tag -- python3 tools/op_bench.py --hid 1 $a --reps 10 > $OUT/pmc_f_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/r02_opt3"
for tag in ("d0", "d3", "d1", "d2", "l0", "l1", "l3"):
    acc = [0, 0.0]
    for f in glob.glob(os.path.join(out, "pmc_f_" + tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == "FETCH_SIZE" and "bsr_mfma" in r["Kernel_Name"]:
                acc[0] += 1; acc[1] += float(r["Counter_Value"])
    print(tag, "FETCH x2 = %.1f MB" % (2 * acc[1] / max(1, acc[0]) * 1024 / 1e6), open(os.path.join(out, "pmc_f_%s.log" % tag)).read().strip()[-60:])
PY
