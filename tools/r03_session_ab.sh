#!/bin/bash
# round 3: blocked Gauss-Jordan inverse (gj_block): tests, setup time on the 1024^2 lattice
OUT=gpurun_out/${1:-r03ab}
mkdir -p $OUT
export OMP_NUM_THREADS=1
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engine.py -x -q -m gpu -k "gauss_jordan or dense_schur_inverse or solved_directly or device_side_setup or adaptive_gpu_setup or three_product" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for blk in 32 64 32; do
  SW_ENGINE_OPTS="gj_block=$blk" timeout -k 10 400 python bench.py --workload synthetic --lattice 1024 --nb 64 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs > $OUT/b1024_gj$blk.json 2> $OUT/b1024_gj$blk.err || { tail -5 $OUT/b1024_gj$blk.err; exit 1; }
done
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs > $OUT/b128.json 2> $OUT/b128.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    c = d["config"]
    sl = (c.get("solver") or {}).get("setup_log") or []
    print("%-22s value %8.1f iters %s setup %.2f s  %s" % (f.split("/")[-1], d["value"], c["outer_iterations_max"], c.get("setup_s") or 0, sl[-1] if sl else ""))
PY
