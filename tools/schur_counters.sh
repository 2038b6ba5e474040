#!/bin/bash
# SQ / cache counters of the even-odd smoother step of the stencil level (k_schur_step) inside the
# default bench workload, one stream; one --pmc pass per counter group (no trace domains with --pmc)
OUT=gpurun_out/r02_schur
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --streams 1 --no-large-stencil --no-cpu-baseline --no-f32-line"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
done
python3 - <<'PY'
import csv, glob, os
out = "gpurun_out/r02_schur"
for d in sorted(glob.glob(out + "/p[0-9]")):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = None
            if "k_schur_step<" in name and "cplxf" not in name and "float" not in name: key = "k_schur_step"
            elif "k_stencil<0" in name: key = "k_stencil<0>"
            elif "k_multidot<4" in name: key = "k_multidot<4>"
            if key is None: continue
            a = acc.setdefault((key, r["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for (k, c), v in sorted(acc.items()):
        print("%-14s %-28s %.5g per launch (%d launches)" % (k, c, v[1] / v[0], v[0]))
PY
