#!/bin/bash
# SQ counters of the level-1 block operator (k_bsr_mfma<3,4>) and the dense kernel, back-to-back
OUT=gpurun_out/r02_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/op_bench.py --hid 1 --level 1 --mode 2 --reps 5 > $OUT/warm.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/op_bench.py --hid 1 --level 1 --mode 2 --reps 10 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/op_bench.py --hid 1 --level 1 --mode 2 --reps 10 > $OUT/p2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/p3 -- python3 tools/op_bench.py --hid 1 --level 1 --mode 2 --reps 10 > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/r02_sq"
for d in ("p1", "p2", "p3"):
    acc = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bsr_mfma" not in r["Kernel_Name"]: continue
            a = acc.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k, "%.4g per launch" % (v[1] / v[0]))
    if not acc:
        print(d, "no data:", open(os.path.join(out, d + ".log")).read()[-400:])
PY
