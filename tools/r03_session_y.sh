#!/bin/bash
# round 3: hardware counters of k_schur_step inside one bench batch (what bounds it: issue, waits, L1/L2 traffic)
OUT=gpurun_out/${1:-r03y}
mkdir -p $OUT
export OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CMD="python3 bench.py --steps 1 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs --engine-opts eo_walk=0"
pass() {  # name, counters...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_schur_step" --output-format csv -d $OUT/$n -- $CMD > $OUT/$n.json 2> $OUT/$n.err || { echo "pass $n failed"; tail -3 $OUT/$n.err; return 1; }
}
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM &&
pass p2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS &&
pass p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum &&
pass p4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_TA_BUSY_sum &&
pass p5 GRBM_GUI_ACTIVE GRBM_COUNT TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
python3 - <<PY
import csv, glob, os
out = "$OUT"
for d in ("p1", "p2", "p3", "p4", "p5"):
    acc = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            mode = k.split("k_schur_step<")[1].split(">")[0] if "k_schur_step<" in k else k[:40]
            a = acc.setdefault((mode, r["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()):
        print(d, k[0][-6:], k[1], "%.5g per launch (%d launches)" % (v[1] / v[0], v[0]))
PY
