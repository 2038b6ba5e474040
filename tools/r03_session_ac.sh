#!/bin/bash
# round 3: run-to-run spread of the device-side setup on the 1024^2 lattice (five processes, one box)
OUT=gpurun_out/${1:-r03ac}
mkdir -p $OUT
export OMP_NUM_THREADS=1
for i in 1 2 3 4 5; do
  timeout -k 10 400 python bench.py --workload synthetic --lattice 1024 --nb 64 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-line --no-large-stencil --no-other-configs > $OUT/b1024_$i.json 2> $OUT/b1024_$i.err || { tail -5 $OUT/b1024_$i.err; exit 1; }
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    c = d["config"]
    sl = (c.get("solver") or {}).get("setup_log") or []
    print("%-22s value %8.1f iters %s setup %.2f s (solver %.2f)  %s" % (f.split("/")[-1], d["value"], c["outer_iterations_max"], c.get("setup_s") or 0, (c.get("solver") or {}).get("setup_s") or 0, sl[-1] if sl else ""))
    print("     ", [ (e.get("pass"), e.get("level"), e.get("seconds")) for e in sl[:-1]])
PY
