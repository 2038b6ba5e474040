#!/bin/bash
# GPU session: PMC calibration of the engine's access shapes + A/B timings and PMC passes of the
# coarse-operator kernels (level-1 block operator on MFMA, level-0 prolongator).
set -e
OUT=gpurun_out/r02_opt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- build/pmc_calibrate > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -- build/pmc_calibrate > $OUT/cal_write.log 2>&1
B="python3 tools/op_bench.py"
{
$B --hid 1 --level 1 --mode 2
$B --hid 1 --level 1 --mode 2 --opt mfma_tiles=8
$B --hid 1 --level 1 --mode 0
$B --hid 1 --level 1 --mode 2 --opt mfma_ops=0
$B --hid 1 --level 0 --what P --opt ell_order=0
$B --hid 1 --level 0 --what P --opt ell_order=1
$B --hid 1 --level 1 --what P --opt ell_order=0
$B --hid 1 --level 1 --what P --opt ell_order=1
$B --hid 1 --level 0 --what R
$B --hid 1 --level 0 --what cinv
$B --hid 1 --level 0 --mode 2
} > $OUT/timings.jsonl 2> $OUT/timings.err
cat $OUT/timings.jsonl
for v in l1op:"--hid 1 --level 1 --mode 2" P0old:"--hid 1 --level 0 --what P --opt ell_order=0" P0new:"--hid 1 --level 0 --what P --opt ell_order=1"; do
  tag=${v%%:*}; a=${v#*:}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f_$tag -- python3 tools/op_bench.py $a --reps 10 > $OUT/pmc_f_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w_$tag -- python3 tools/op_bench.py $a --reps 10 > $OUT/pmc_w_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/r02_opt"
def avg(dirname, counter):
    acc = {}
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter: continue
            k = r["Kernel_Name"].split("(")[0][:60]
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] * 1024 / 1e6 for k, v in acc.items()}
lines = []
for d, c in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
    for k, v in sorted(avg(d, c).items()):
        lines.append("%-12s %-62s %10.1f MB raw (touched 1073.7 MB)" % (c, k, v))
for tag in ("l1op", "P0old", "P0new"):
    for d, c in (("pmc_f_" + tag, "FETCH_SIZE"), ("pmc_w_" + tag, "WRITE_SIZE")):
        for k, v in sorted(avg(d, c).items()):
            if "bsr" in k or "k_ell" in k:
                lines.append("%-8s %-12s %-62s %10.1f MB raw" % (tag, c, k, v))
open(os.path.join(out, "pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
