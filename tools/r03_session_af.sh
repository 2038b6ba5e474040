#!/bin/bash
# round 3: why k_schur_step runs at 4.7 of 8 waves per SIMD: workgroup dispatch / resource counters
OUT=gpurun_out/${1:-r03af}
mkdir -p $OUT
export OMP_NUM_THREADS=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 1 --warmup 1 --no-large-stencil --no-cpu-baseline --no-f32-line --no-other-configs --engine-opts eo_lds=0"
pass() {  # name, counters...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "k_schur_step" --output-format csv -d $OUT/$n -- $CMD > $OUT/$n.json 2> $OUT/$n.err || { echo "pass $n failed"; tail -3 $OUT/$n.err; return 1; }
}
pass p1 SPI_CSN_BUSY SPI_CSN_WINDOW_VALID SPI_CSN_NUM_THREADGROUPS SPI_CSN_WAVE SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN &&
pass p2 SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN SPI_RA_LDS_CU_FULL_CSN &&
pass p3 MeanOccupancyPerCU MeanOccupancyPerActiveCU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE &&
pass p4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAVES
python3 - <<PY
import csv, glob, os
out = "$OUT"
for d in ("p1", "p2", "p3", "p4"):
    acc = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()):
        print(d, k, "%.5g per launch (%d launches)" % (v[1] / v[0], v[0]))
PY
