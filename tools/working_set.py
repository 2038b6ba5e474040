#!/usr/bin/env python3
"""Back-to-back stencil rate against the working set (batch width) on one lattice: how much of
the 128^2 rate is the 256 MB Infinity Cache.  Mode 0: Y = A X (two vectors ping-pong), mode 2:
fused smoother step (three vectors)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deflatedmlmc_schwinger_amd import matrix as swm  # noqa: E402
from deflatedmlmc_schwinger_amd.engine import Engine  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
U1, U2 = swm.synthetic_links(L, 0.45, 2024)
for nb in (64, 128, 256, 512, 1024):
    for mode in (0, 2):
        eng = Engine(0)
        eng.hier_begin(0, 1)
        eng.set_lattice(0, L, -0.05, U1, U2)
        eng.hier_end(0)
        eng.set_option("bench_mode", mode)
        ms = eng.bench_dirac(0, 0, nb, 40)
        work = L * L * ((64.0 if mode == 0 else 96.0) * nb + 32.0)
        vecs = 2 if mode == 0 else 3
        print(json.dumps({"L": L, "nb": nb, "mode": mode, "working_set_MB": round(vecs * 2 * L * L * nb * 16 / 1e6, 1),
                          "us": round(ms * 1e3, 2), "GBs": round(work / ms / 1e6, 1)}), flush=True)
        eng.close()
