#!/usr/bin/env python3
"""Stencil HBM-roofline sweep: lattice size x batch x non-temporal stores (back-to-back launches)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deflatedmlmc_schwinger_amd import matrix as swm
from deflatedmlmc_schwinger_amd.engine import Engine
for L, nb in ((128, 256), (512, 64), (1024, 64), (2048, 32)):
    U1, U2 = swm.synthetic_links(L, 0.45, 2024)
    for nt in (0, 1):
        eng = Engine(0)
        eng.hier_begin(0, 1); eng.set_lattice(0, L, -0.05, U1, U2); eng.hier_end(0)
        eng.set_option("stencil_nt", nt)
        ms = eng.bench_dirac(0, 0, nb, 20)
        nbp = ((nb + 63) // 64) * 64
        work = L * L * (64.0 * nbp + 32.0)
        print(json.dumps({"L": L, "nb": nb, "nt_store": nt, "ms": round(ms, 4),
                          "GBs": round(work / ms / 1e6, 1), "frac": round(work / ms / 1e6 / 8000, 3)}), flush=True)
        eng.close()
