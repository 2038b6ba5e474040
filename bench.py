#!/usr/bin/env python3
"""Headline benchmark: Hutchinson probe-samples/s on schwinger128 (BASELINE.json).

One step = one batch of NB (default 256) deflated-Hutchinson probes per engine stream per GPU:
the probes are GENERATED on the GPU inside the timed region (the reference's MT19937 stream,
bit-exact, k_mt_generate) and solved as one multi-RHS batch in fp64 to a relative residual of
1e-12 (config 2 of BASELINE.json; with --gpus N the probe stream is sharded in contiguous
blocks over N ranks, config 4, every rank jumping straight to its stream positions).  The line
printed by rank 0 also carries

* ``value_probes_resident``: the same steps with the probes already resident in HBM (round 1's
  headline), and ``value_pcie_inclusive_this_rank``: with host-made probes uploaded per step;

* ``roofline``: achieved algorithmic HBM rate of the batched Wilson stencil kernel, from
  HIP-event timings of that kernel on the engine's stream in an instrumented extra step of
  the same workload (kept out of the timed region so the events do not perturb ``value``);
* ``cpu_baseline``: the oracle's reference-faithful NumPy/SciPy path timed on this host
  (rank 0, N = 1 only, a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6     # MI355X datasheet FP64 matrix (dense)
MFMA_F64_MEASURED_ISSUE_TFLOPS = 48.1   # pure-issue rate of v_mfma_f64_16x16x4, 4 waves per SIMD (profiles/r01_mfma_f64_rate.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults = what the round driver passes: with 4 steps after ONE warm-up batch the first timed batches still
    # carry first-use costs -- 7.7 ms per step against 7.1-7.2 over 20 steps after 5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nb", type=int, default=256, help="probes per engine stream per step")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SW_STREAMS", "0")),
                    help="concurrent probe batches (engine handles / HIP streams) per GPU; one since round "
                         "3: with the block level solved directly a single batch keeps the GPU busy and "
                         "its smoother's working set (201 MB) inside the 256 MB Infinity Cache, which three "
                         "concurrent batches thrash (29.8k / 27.8k / 27.5k probe-samples/s for 1 / 2 / 3); "
                         "the same holds for --workload mlmc now that its coarse-level solves are direct "
                         "(33.6k / 31.6k / 29.9k).  0: that default")
    ap.add_argument("--tol", type=float, default=1e-12)
    ap.add_argument("--stop-factor", type=float, default=float(os.environ.get("SW_STOP_FACTOR", "0.1")),
                    help="every batch is iterated until its TRUE residuals are below stop_factor * tol (iteration "
                         "counts are still reported at tol).  0.1 (default, the headline): the north star's "
                         "per-probe criterion as written -- every one of the 256 estimates of a batch within 1e-10 "
                         "relative of the LU oracle, near-cancelling estimates included "
                         "(tests/test_gpu_golden.py); 1.0: the reference's own stopping point, reported beside it "
                         "as value_reference_stopping_point")
    ap.add_argument("--cfg", type=str, default=os.environ.get("SW_SOLVER_CFG", ""),
                    help="JSON solver-hierarchy override")
    ap.add_argument("--lattice", type=int, default=1024, help="extent of the synthetic lattice "
                    "(--workload synthetic only)")
    ap.add_argument("--workload", choices=["hutchinson", "mlmc", "synthetic", "config2"],
                    default="hutchinson",
                    help="hutchinson: deflated Hutchinson probes (BASELINE configs 2/4, the "
                         "headline metric); mlmc: level-0 MLMC difference probes with level "
                         "skipping, A0^-1 - P0 P1 A2^-1 R1 R0 (config 3); synthetic: plain Hutchinson "
                         "probes on a synthetic random-gauge --lattice^2 configuration with the "
                         "GPU-side setup (config 5; use --nb 64 and --streams 1 or 2 at 1024); "
                         "config2: BASELINE config 2 AS WRITTEN -- plain (k=0) Hutchinson probes, "
                         "2-level multigrid 32768 -> 8192 built with the reference's aggregation "
                         "(32-row aggregates, 4 test vectors x 2), dense 8192^2 coarse inverse")
    ap.add_argument("--synthetic-levels", type=int, default=0,
                    help="--workload synthetic: 0 the tuned hierarchy (five levels at 1024^2), 3 BASELINE config 5 as "
                         "written (2097152 -> 262144 -> 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-line", action="store_true",
                    help="skip the secondary measurement with the single-precision preconditioner")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact driver-run records of the other BASELINE configurations "
                         "(config 2 as written, config 3's MLMC difference, config 5's 1024^2 lattice, the "
                         "strict parity mode, the drop-in G102 / G202 flows)")
    ap.add_argument("--no-large-stencil", action="store_true",
                    help="skip the synthetic 1024^2 stencil roofline point")
    ap.add_argument("--engine-opts", type=str, default=os.environ.get("SW_ENGINE_OPTS", ""),
                    help="comma-separated name=value engine switches (sw_set_option) for A/B runs")
    ap.add_argument("--cpu-probes", type=int, default=8)
    ap.add_argument("--cpu-workers", type=int, default=-1,
                    help="processes of the all-cores CPU baseline line (-1: min(cores, 16); 0: skip)")
    return ap.parse_args()


def first_probe(step, world, rank, streams, stream, nb):
    """Stream position (in probes) of the block that engine `stream` of rank `rank` works on in round
    `step`: a round is world * streams batches of nb consecutive probes of the MT19937(123456) stream,
    one contiguous block per (rank, stream) -- BASELINE config 4's sharding, at the single-GPU order's
    stream positions."""
    return ((step * world + rank) * streams + stream) * nb


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes
    (before this process touches the GPU) and leave with their exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.streams <= 0:
        args.streams = 1
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s; refusing to report a line for a "
                         "different rank count\n" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
        sys.exit(2)
    # Libraries (RCCL's version banner at communicator creation, for one) write to the process's
    # stdout; the contract is ONE JSON line there.  Everything written to fd 1 during the run is
    # sent to stderr, the JSON line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
    if line is not None:
        os.write(1, (line + "\n").encode())
    os.close(real_stdout)


def build_problem(workload, args, device_index, engines):
    """Operands, hierarchies and deflation vectors of one workload (untimed setup).
    Returns (mg, A, trace_params, tr1, unknowns, setup_seconds)."""
    import contextlib
    import io
    from deflatedmlmc_schwinger_amd import gateway, matrix, utils
    from deflatedmlmc_schwinger_amd import hierarchy as swhier
    from deflatedmlmc_schwinger_amd.multigrid import MG
    params = gateway.set_params('schwinger128')
    params['function_tol'] = args.tol
    params['device'] = device_index
    params['engines'] = engines
    params['stop_factor'] = args.stop_factor
    if args.cfg and workload == args.workload:
        params['solver_cfg'] = json.loads(args.cfg)
    elif workload in ("hutchinson", "mlmc"):
        params['solver_cfg'] = dict(swhier.TUNED_SOLVER_CFG_128)
    t_setup = time.time()
    if workload == "synthetic":
        # BASELINE config 5: sigma = 0.204 <-> mean plaquette ~0.92, m = -0.05
        Ls = args.lattice
        U1s, U2s = matrix.synthetic_links(Ls, 0.204, 2024)
        # hierarchy.synthetic_solver_cfg: 8x8 site aggregates once, then 2x2 down to 16 x 16 sites, every
        # level smoothed even-odd, a 2-step K-cycle on level 1 (profiles/r02_synthetic_lattices.txt,
        # 1024^2: 408 probe-samples/s, 9 iterations; Schur steps on the lattice level: 10 -- 512^2: 1377
        # probe-samples/s against 1331 with 14, 1024^2: 408 against 386, one more outer iteration)
        scfg = swhier.synthetic_solver_cfg(Ls, int(os.environ.get("SW_SYNTH_NU0", "10")),
                                           os.environ.get("SW_SYNTH_SETUP", "device"),
                                           levels=getattr(args, "synthetic_levels", None))
        if args.cfg and workload == args.workload:
            scfg = json.loads(args.cfg)
        mg = MG((Ls, -0.05, U1s, U2s))
        with contextlib.redirect_stdout(io.StringIO()):
            mg.setup_solver_only(scfg, device=device_index, engines=engines)
        for e_ in mg.engines:
            e_.set_option("stop_factor", args.stop_factor)
        return mg, None, None, 0.0, 2 * Ls * Ls, time.time() - t_setup
    if workload == "config2":
        # BASELINE config 2 as written: plain Hutchinson, 2-level MG 32768 -> 8192 from the
        # reference's own aggregation, dense coarse inverse; no solver hierarchy, no deflation
        params['max_nr_levels'] = 2
        params['nr_deflat_vctrs'] = 0
        params['use_solver_hierarchy'] = False
        # smoother of the lattice level: a degree-48 polynomial on its even-odd Schur complement (half vectors,
        # product form, outer solve on the reduced system): 14 iterations against 22 with a degree-48
        # polynomial on the full operator (profiles/r03_ab_sessions.txt, r03s, r03aj)
        params['ref_smoother'] = os.environ.get("SW_CONFIG2_SMOOTHER", "eo")
        params['ref_cycle_post'] = int(os.environ.get("SW_CONFIG2_NU", "48"))
        params['solver_restart'] = int(os.environ.get("SW_CONFIG2_RESTART", "16"))
        # the 8192-row coarse level solved exactly in even-odd form (dense 4096^2 inverse of the Schur
        # complement of its 16-row tiles + three sparse block operators) instead of the dense 8192^2 inverse
        params['ref_coarsest'] = os.environ.get("SW_CONFIG2_COARSEST", "eo")
    A = matrix.loadMatrix(params['matrix'], params['matrix_params'])
    tp = utils.trace_params_from_params(params, "mlmc" if workload == "mlmc" else "hutchinson")
    mg = MG(A)
    with contextlib.redirect_stdout(io.StringIO()):
        mg.setup(dof=tp['dof'], aggrs=tp['aggrs'], max_levels=tp['max_nr_levels'], dim=2,
                 acc_eigvs=tp['accuracy_mg_eigvs'], sys_type=tp['problem_name'], params=tp)
        Ux, tr1 = utils.deflation_pre_computations(A, tp['nr_deflat_vctrs'],
                                                   tp['defl_eigvs_tol_Hutch'], "hutchinson",
                                                   mg.timer, tp, mg)
    return mg, A, tp, tr1, A.shape[0], time.time() - t_setup


KERNEL_CLASSES = (("k_stencil<0>", 8), ("k_stencil<1>", 9), ("k_stencil<2>", 10),
                  ("k_bsr_mfma(dense coarsest)", 11), ("k_bsr_mfma(level-1 operator)", 12),
                  ("k_bsr_mfma(level-2 operator)", 14), ("k_schur_step", 15),
                  ("k_schur_step<0/1> (S x, b' - S x)", 16))
MFMA_CLASSES = (("k_bsr_mfma(dense coarsest)", 11), ("k_bsr_mfma(level-1 operator)", 12),
                ("k_bsr_mfma(level-2 operator)", 14))
# classes whose algorithmic work the engine counts launch by launch while profiling: the MFMA classes
# (complex multiply-adds x 8 flops) and the even-odd smoother of the lattice level (bytes: its launches
# cover row windows of different heights in the time-skewed order)
COUNTED_CLASSES = MFMA_CLASSES + (("k_schur_step", 15),)


def kernel_rooflines(eng, levels, V, nbp, pmc=None, skip=(), three_products=True):
    """Per-kernel-class roofline records from the engine's HIP-event buckets of an instrumented batch
    (profiling on): achieved = algorithmic bytes (or issued flops) per launch / average launch time
    (DESIGN.md section 5), sorted by the class's share of the batch."""
    kstats = {name: eng.kernel_stats(cls) for name, cls in KERNEL_CLASSES if name not in skip}
    kwork = {name: eng.kernel_work(cls) for name, cls in COUNTED_CLASSES}
    nc = levels[-1]
    algo = {
        "k_stencil<0>": ("hbm", V * (64.0 * nbp + 32.0)),
        "k_stencil<1>": ("hbm", V * (96.0 * nbp + 32.0)),              # + read of B
        "k_stencil<2>": ("hbm", V * (96.0 * nbp + 32.0)),              # fused smoother step
        # even-odd smoother step / hop: three HALF-vector passes (x_e, b'_e in, x_e out) + links
        "k_schur_step": ("hbm", 0.5 * V * (96.0 * nbp + 64.0)),
        # operator of the even-odd reduced system (outer Krylov solver on half vectors): two
        # half-vector passes (the few residual launches, three, are counted at the same figure)
        "k_schur_step<0/1> (S x, b' - S x)": ("hbm", 0.5 * V * (64.0 * nbp + 64.0)),
        "k_bsr_mfma(dense coarsest)": ("mfma", 8.0 * nc * nc * nbp),
        "k_bsr_mfma(level-1 operator)": ("mfma", 8.0 * levels[1] * 80.0 * nbp if len(levels) > 2 else 0.0),
        # (with more than one such level the class averages over them; the 128^2 hierarchies have one)
        "k_bsr_mfma(level-2 operator)": ("mfma", 8.0 * levels[2] * 80.0 * nbp if len(levels) > 3 else 0.0),
    }
    peaks = {"hbm": (HBM_PEAK_GBS, "GB/s", 1e9), "mfma": (MFMA_F64_PEAK_TFLOPS, "TFLOP/s", 1e12)}
    pmc = pmc or {}
    out = []
    for name, (ms_tot, cnt) in kstats.items():
        if cnt == 0:
            continue
        bound, work = algo[name]
        if name in kwork and kwork[name] > 0.0:
            # MFMA classes: the complex multiply-adds the launches performed, at 8 real flops each (full
            # operator, even-odd Schur steps, dense inverses have different shapes), counted by the
            # engine.  (The three-product kernel issues 6 real flops per complex multiply-add: `achieved`
            # is the ALGORITHMIC rate, and may approach 4/3 of what the matrix cores execute.)
            work = kwork[name] / cnt
        peak, unit, scale = peaks[bound]
        avg_ms = ms_tot / cnt
        ach = work / (avg_ms * 1e-3) / scale
        rec = {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
               "frac": ach / peak, "traffic": pmc.get(name, {}).get("hbm_bytes_per_launch"),
               "work_per_launch": work, "avg_launch_ms": avg_ms, "launches_in_step": cnt, "step_ms": ms_tot}
        if rec["traffic"] is not None:
            rec["traffic_source"] = "profiles/kernel_pmc.json (rocprofv3 --pmc passes of this workload, builder-run)"
        if name == "k_schur_step" and cnt > 0:
            # The launches of this class move 3 half-vector passes (steps, the residual and the fused last
            # factor of the product form) or 2 (its middle factors), and a 2-pass launch takes 0.92 of the
            # time of a 3-pass one: the kernel's time is a wave's latency chain, not its HBM bytes (DESIGN
            # section 4 item 3, profiles/r03_ab_sessions.txt r03w-r03ag).  Beside the algorithmic HBM rate,
            # the rate of row requests to the L2: 18 neighbour rows + its own 2 (+ 2 of b') + 2 stores per
            # output site, 22 or 20 KiB-rows per site and 64-probe chunk, 18 of them L2 hits.
            b3, b2 = 0.5 * V * (96.0 * nbp + 64.0), 0.5 * V * (64.0 * nbp + 64.0)
            n3 = min(float(cnt), max(0.0, (work * cnt - cnt * b2) / (b3 - b2)))
            l2_bytes = (n3 * 22.0 + (cnt - n3) * 20.0) * 0.5 * V * 16.0 * nbp / cnt
            rec["l2_side"] = {"rows_requested_per_site": "22 (3-pass launches) / 20 (2-pass)",
                              "three_pass_launches": int(round(n3)), "two_pass_launches": int(round(cnt - n3)),
                              "bytes_through_l2_per_launch": l2_bytes,
                              "achieved_TBs": l2_bytes / (avg_ms * 1e-3) / 1e12,
                              "guide_ceiling_TBs": "16.8-18.8 (L2-served gathers, MI355X_MICROARCH.md)",
                              "note": "full-lattice launches only (the time-skewed strips of larger lattices "
                                      "are counted by their own rows)"}
        if bound == "hbm":
            # what "hbm" means at this size: the vectors a smoother sequence / an operator application touches
            # (three / two half vectors, two full ones for the stencil) against the 256 MB Infinity Cache --
            # below it the rate is a cache-resident (fabric-side) rate, FETCH_SIZE / WRITE_SIZE count
            # Infinity-Cache hits too (MI355X_MICROARCH.md)
            ws = {"k_schur_step": 3 * 0.5, "k_schur_step<0/1> (S x, b' - S x)": 2 * 0.5}.get(name, 2.0) \
                * V * 32.0 * nbp
            rec["working_set_MB"] = ws / 1e6
            rec["bound_detail"] = ("hbm+infinity-cache (working set %.0f MB <= 256 MB: cache-resident rate)"
                                   % (ws / 1e6)) if ws <= 256e6 else "hbm (working set %.0f MB)" % (ws / 1e6)
        if bound == "mfma" and three_products:
            # the three-product kernels execute 6 real flops per complex multiply-add, `achieved` counts 8
            rec["executed_frac_of_peak"] = rec["frac"] * 0.75
            # what v_mfma_f64_16x16x4 sustains on this chip when NOTHING but it is issued (tools/mfma_f64_rate.hip,
            # profiles/r01_mfma_f64_rate.txt: 35.8 / 47.4 / 48.1 TFLOP/s at 1 / 2 / 4 waves per SIMD -- 105
            # 2.4-GHz cycles per instruction against the datasheet's 64): the executed rate against THAT ceiling
            rec["measured_issue_ceiling_TFLOPs"] = MFMA_F64_MEASURED_ISSUE_TFLOPS
            rec["executed_frac_of_measured_issue_ceiling"] = (ach * 0.75) / MFMA_F64_MEASURED_ISSUE_TFLOPS
        out.append(rec)
    out.sort(key=lambda r: -r["step_ms"])
    return out


def secondary(label, mg, run_mode, nb, n, tol, steps, warmup, engine_opts=None):
    """A compact driver-run record of another BASELINE configuration on ONE engine / one stream:
    `steps` timed batches (probes generated on the device inside the timed region, next batch
    prefetched), then one instrumented batch for the dominant kernel and its roofline fraction."""
    import torch
    from deflatedmlmc_schwinger_amd.engine import ProbeStream
    eng = mg.engine
    saved_opts = {k: eng.get_option(k) for k in (engine_opts or {})}
    for k, v in (engine_opts or {}).items():
        eng.set_option(k, v)
    eng.stream_set(ProbeStream(123456).window())
    its = []

    def one(s, prefetch):
        slot = s & 1
        if s == 0 or not one.ready:
            eng.probes_generate(slot, 0, nb, s * nb * n)
        one.ready = False
        if prefetch:
            eng.probes_generate(1 - slot, 0, nb, (s + 1) * nb * n)
            one.ready = True
        eng.probes_select(slot)
        eng.hutch_run(run_mode, 0, tol, 1000)
        e, itf, _ = eng.hutch_fetch()
        its.append(int(itf.max()))
        return e
    one.ready = False
    for s in range(warmup):
        one(s, False)
    one.ready = False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        one(s, s < warmup + steps - 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.set_profiling(True)
    eng.timers_reset()
    one.ready = False
    one(warmup + steps, False)
    levels = mg.solver_info["levels"] if mg.solver_info else [lev.A.shape[0] for lev in mg.ml.levels]
    L = mg.lattice[0]
    nbp = ((nb + 63) // 64) * 64
    roofs = kernel_rooflines(eng, levels, L * L, nbp,
                             skip=("k_bsr_mfma(level-1 operator)", "k_bsr_mfma(level-2 operator)")
                             if len(levels) > 3 and mg.solver_info else ())
    buckets = eng.timers()
    launches = eng.launch_count()
    eng.set_profiling(False)
    for k, v in saved_opts.items():
        eng.set_option(k, v)
    dom = roofs[0] if roofs else None
    return {"workload": label, "value": steps * nb / dt, "unit": "probe-samples/s", "steps": steps,
            "probes_per_step": nb, "ms_per_step": 1e3 * dt / steps,
            "outer_iterations_max": max(its[warmup:warmup + steps]),
            "levels": levels, "kernel_launches_per_batch": launches,
            "step_breakdown_ms": buckets,
            "dominant_kernel": None if dom is None else
            {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                 "launches_in_step", "step_ms")}}


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as td
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path")
    # one process per GPU; SW_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
    backend = os.environ.get("SW_DIST_BACKEND", "nccl")
    device_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    if world > 1 or os.environ.get("SW_FORCE_PROCESS_GROUP"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        td.init_process_group(backend=backend, rank=rank, world_size=world)

    line_out = None
    from deflatedmlmc_schwinger_amd import dist as swdist
    from deflatedmlmc_schwinger_amd import gateway, matrix, utils
    from deflatedmlmc_schwinger_amd.engine import MODE_HUTCHINSON, MODE_MLMC_SKIP, ProbeStream
    from deflatedmlmc_schwinger_amd.multigrid import MG, REF_HID

    # ---- setup (untimed): operands, hierarchies, deflation vectors --------------------
    from deflatedmlmc_schwinger_amd import hierarchy as swhier
    synthetic = args.workload == "synthetic"
    run_mode = MODE_MLMC_SKIP if args.workload == "mlmc" else MODE_HUTCHINSON
    mg, A, tp, tr1, n_unknowns, t_setup = build_problem(args.workload, args, device_index, max(1, args.streams))
    eng = mg.engine
    n = n_unknowns
    L = mg.lattice[0]
    V = L * L
    nb = args.nb
    nbp = ((nb + 63) // 64) * 64
    maxiter = 1000

    # ---- the probe stream: MT19937(123456) as stoch_trace.py:103 seeds it; every engine holds
    # the window at stream position 0 and jumps to its own block of every round -------------------
    engs = mg.engines
    ne = len(engs)
    for kv in [x for x in args.engine_opts.split(",") if x]:
        for e_ in engs:
            e_.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    window0 = ProbeStream(123456).window()
    for e in range(ne):
        engs[e].stream_set(window0)
    if td.is_initialized() and os.environ.get("SW_ENGINE_COMM") == "1" and backend == "nccl":
        # the statistics all-reduce through the engine's own C-ABI collective (sw_allreduce_stats)
        comm = swdist.EngineComm(engs[0])
    else:
        comm = swdist.TorchComm() if td.is_initialized() else swdist.Comm()
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=ne)

    def block_start(s, e):
        return first_probe(s, world, rank, ne, e, nb)

    gen_ready = [None] * ne      # step whose probes engine e already holds in slot (step & 1)

    def run_one(e, s, generate=True, slot=0, prefetch=False):
        """One batch on engine e.  generate: its probes are drawn on the device, into slot (s & 1) --
        unless the previous step already queued them there (prefetch: the engine's generation stream
        draws step s + 1's probes while step s is being solved; EXACTLY one generation per timed step
        stays inside the timed region, see headline())."""
        if generate:
            slot = s & 1
            if gen_ready[e] != s:
                engs[e].probes_generate(slot, 0, nb, block_start(s, e) * n)
            if prefetch:
                engs[e].probes_generate(1 - slot, 0, nb, block_start(s + 1, e) * n)
                gen_ready[e] = s + 1
        engs[e].probes_select(slot)
        engs[e].hutch_run(run_mode, 0, args.tol, maxiter)
        return engs[e].hutch_fetch()

    def step(s, generate=True, slot=0, prefetch=False):
        if ne == 1:
            res = [run_one(0, s, generate, slot, prefetch)]
        else:
            res = list(pool.map(lambda e: run_one(e, s, generate, slot, prefetch), range(ne)))
        ests = np.concatenate([r[0] for r in res])
        itf = np.concatenate([r[1] for r in res])
        # this rank's {sum Re e, sum Im e, sum |e|^2, n}: accumulated over the steps and all-reduced ONCE
        # per timed region (the path's single collective) -- not once per step
        return ests, itf, swdist.local_stats(ests)

    def timed(fn, ends_with_collective=False):
        """Barrier + device synchronisation, the clock, fn(), closing synchronisation, the clock; MAX over the
        ranks.  ends_with_collective: fn's last action is the path's own blocking all-reduce of the
        statistics -- every rank leaves it only after every rank has entered it, i.e. it IS the closing
        barrier, and a second one would only add its fixed cost to every N > 1 point of the scaling curve."""
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if not ends_with_collective:
            comm.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if td.is_initialized():
            tmax = torch.tensor([dt], dtype=torch.float64,
                                device="cuda" if backend == "nccl" else "cpu")
            td.all_reduce(tmax, op=td.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # ---- the timed region: K steps of generate + solve + reduce ---------------------------------
    for s in range(args.warmup):
        step(s)
    comm.allreduce_stats(np.zeros(4))     # warm-up of the collective itself (lazy communicator set-up)
    total = np.zeros(4)
    iters_seen = []

    def headline():
        last = args.warmup + args.steps - 1
        for s in range(args.warmup, last + 1):
            # step s + 1's probes are drawn during step s; the first timed step draws its own (the
            # warm-up did not prefetch them) and the last one prefetches nothing: K generations in K steps
            ests, itf, stats = step(s, prefetch=(s < last))
            total[:] += stats
            iters_seen.append(int(itf.max()))
        # the one collective of the path: trace sum / variance statistics over the ranks (RCCL over xGMI)
        total[:] = comm.allreduce_stats(total)
    elapsed = timed(headline, ends_with_collective=True)

    # ---- the same K steps at the reference's own stopping point (stop_factor = 1: residual below tol, where
    # pyamg's fgmres stops; per-probe parity then holds to 1e-10 of max(|e|, a tenth of the batch median)) ----
    elapsed_refstop = None
    if args.stop_factor != 1.0:
        for e_ in engs:
            e_.set_option("stop_factor", 1.0)
        for e in range(ne):
            gen_ready[e] = None
        step(max(0, args.warmup - 1))      # one untimed step in the other mode

        def refstop():
            acc = np.zeros(4)
            last = args.warmup + args.steps - 1
            for s in range(args.warmup, last + 1):
                acc += step(s, prefetch=(s < last))[2]
            comm.allreduce_stats(acc)
        for e in range(ne):
            gen_ready[e] = None
        elapsed_refstop = timed(refstop, ends_with_collective=True)
        for e_ in engs:
            e_.set_option("stop_factor", args.stop_factor)
        for e in range(ne):
            gen_ready[e] = None

    # ---- secondary: the same steps with the probes resident in HBM before the clock starts
    # (round 1's definition), and with host-made probes uploaded per step (PCIe-inclusive) -------
    nres = min(args.steps, 8)
    for i in range(nres):
        for e in range(ne):
            engs[e].probes_generate(2 + i, 0, nb, block_start(args.warmup + i, e) * n)
    for e in range(ne):
        engs[e].sync()
    elapsed_resident = timed(lambda: [step(args.warmup + i, False, 2 + i) for i in range(nres)])
    host_batches = [ProbeStream(7 + e).rademacher(nb, n) for e in range(ne)]

    def pcie():
        for i in range(nres):
            for e in range(ne):
                engs[e].probes_upload_slot(2 + i, 0, host_batches[e])
            step(args.warmup + i, False, 2 + i)
    elapsed_pcie = timed(pcie)

    # ---- secondary: the same K steps with the multigrid cycle in single precision (complex64 on the
    # f32 matrix cores) INSIDE the fp64 flexible GMRES -- residuals, orthogonalisation and the
    # true-residual verification stay fp64, every probe converges to the same tol.  Reported next
    # to `value`, never as `value` (which is the all-fp64 path).
    f32_line = None
    scfg_used = (mg.solver_info or {}).get("cfg") if mg.solver_info else None
    if scfg_used and swhier.f32_capable(scfg_used) and not args.no_f32_line:
        for e_ in engs:
            e_.set_option("precond_f32", 1)
        step(args.warmup + args.steps)                      # builds the complex64 mirrors
        its32 = []
        ref_chk = []

        def f32_steps():
            acc = np.zeros(4)
            for s in range(args.warmup, args.warmup + args.steps):
                ests, itf, stats = step(s)
                its32.append(int(itf.max()))
                acc += stats
            ref_chk.append(comm.allreduce_stats(acc))
        elapsed_f32 = timed(f32_steps)
        for e_ in engs:
            e_.set_option("precond_f32", 0)
        tot32 = np.sum(ref_chk, axis=0)
        f32_line = {"value": world * ne * args.steps * nb / elapsed_f32,
                    "ms_per_step": 1e3 * elapsed_f32 / args.steps,
                    "outer_iterations_max": max(its32),
                    # same probes as the fp64 steps: the sums of the per-probe estimates agree to
                    # the solver tolerance
                    "rel_diff_of_estimate_sum_vs_fp64": float(abs(tot32[0] - total[0]) / abs(total[0]))
                    if total[0] != 0 else None,
                    "note": "multigrid preconditioner in complex64 (k_bsr_mfma_f32, k_schur_step<float2>) "
                            "inside the fp64 FGMRES; tol %.0e reached by every probe in fp64" % args.tol}

    # ---- generation alone (device time of k_mt_jump + k_mt_generate per batch) ------------------
    def gen_only():
        for i in range(nres):
            engs[0].probes_generate(0, 0, nb, block_start(args.warmup + i, 0) * n)
        engs[0].sync()
    gen_ms = 1e3 * timed(gen_only) / nres

    # ---- instrumented step: HIP events around every launch, on the engine stream ----------
    eng.set_profiling(True)
    eng.timers_reset()
    run_one(0, args.warmup + args.steps)      # one stream alone: clean per-kernel durations
    levels = mg.solver_info["levels"] if mg.solver_info else [lev.A.shape[0] for lev in mg.ml.levels]
    pmc_all = {}
    try:
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", "kernel_pmc.json")))
    except Exception:
        pmc_all = {}
    if synthetic:
        pmc_all = {}
    elif args.workload != "hutchinson":
        pmc_all = {k: v for k, v in pmc_all.items() if k.startswith(("k_stencil", "k_schur"))}
    kroofs = kernel_rooflines(eng, levels, V, nbp, pmc_all,
                              skip=("k_bsr_mfma(level-1 operator)", "k_bsr_mfma(level-2 operator)")
                              if synthetic else (), three_products="mfma_3m=0" not in args.engine_opts)
    buckets = eng.timers()
    launches = eng.launch_count()
    eng.set_profiling(False)
    # back-to-back stencil launches (no other kernels in between), same buffers
    dirac_ms = eng.bench_dirac(REF_HID, 0, nb, 50)

    if rank == 0:
        mean, std = swdist.mean_and_population_std(total)
        rooflines = kroofs
        dominant = rooflines[0] if rooflines else None
        stencil = max((r for r in rooflines if r["kernel"].startswith(("k_stencil", "k_schur"))),
                      key=lambda r: r["step_ms"], default=None)
        bytes0 = V * (64.0 * nbp + 32.0)      # Y = A X, one launch
        parity_txt = ("[STRICT per-probe parity: stop_factor %.2g, every true residual below %.2g x tol] "
                      % (args.stop_factor, args.stop_factor) if args.stop_factor < 1.0 else
                      "[stopped at the reference's own point: residual below tol] ")
        out = {
            "metric": {"hutchinson": "hutchinson_probe_samples_per_sec_schwinger128",
                       "config2": "hutchinson_probe_samples_per_sec_schwinger128",
                       "mlmc": "mlmc_level0_difference_probe_samples_per_sec_schwinger128",
                       "synthetic": "hutchinson_probe_samples_per_sec_synthetic%d" % L}[args.workload],
            "value": world * ne * args.steps * nb / elapsed,
            "unit": "probe-samples/s",
            # which per-probe parity criterion `value` is measured under
            "parity_mode": ("strict: every true residual below %.2g x tol, all 256 estimates of a batch within "
                            "1e-10 relative of the LU oracle (north_star as written)" % args.stop_factor)
            if args.stop_factor < 1.0 else
            "reference stopping point: residual below tol (1e-10 relative to max(|e|, 0.1 median|e|))",
            "stop_factor": args.stop_factor,
            # the same K steps stopped where the reference's solver stops (stop_factor = 1)
            "value_reference_stopping_point": (world * ne * args.steps * nb / elapsed_refstop)
            if elapsed_refstop else None,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "value_probes_resident": world * ne * nres * nb / elapsed_resident,
            "value_pcie_inclusive_this_rank": ne * nres * nb / elapsed_pcie,
            "probe_generation_ms_per_batch": gen_ms,
            "f32_preconditioner": f32_line,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic Rademacher probes (the reference's MT19937 stream, seed 123456, "
                    "generated on the GPU inside the timed region) on the schwinger128 gauge "
                    "configuration (link fixture), m0=-0.1320" if not synthetic else
                    "synthetic Rademacher probes (MT19937 seed 123456, generated on the GPU inside "
                    "the timed region) on a synthetic random U(1) gauge configuration (seed 2024)",
            "config": {
                "workload": parity_txt + "schwinger128, %d x %d probes/GPU/step (%d multi-RHS batch(es) of %d at a time, "
                            "one HIP stream each), deflated Hutchinson (k=8, Pperm shift 512), tuned solver "
                            "hierarchy %s built on the GPU%s, fp64, tol %.0e"
                            % (ne, nb, ne, nb, "/".join(str(v) for v in levels),
                               (" -- a TWO-LEVEL cycle in effect: the lattice level smoothed even-odd, the "
                                "%d-row level solved exactly (dense inverse of its even-odd Schur complement); "
                                "the last level serves the setup only --" % levels[1])
                               if ((mg.solver_info or {}).get("cfg") or {}).get("direct_levels") == [1] else "",
                               args.tol)
                            if args.workload == "hutchinson" else
                            "BASELINE config 2 as written: schwinger128, %d x %d probes/GPU/step as "
                            "multi-RHS batches, plain Hutchinson (k=0, Pperm shift 512), 2-level "
                            "multigrid 32768 -> 8192 (reference aggregation: 32-row aggregates, 4 test "
                            "vectors x 2; coarse level solved exactly on fp64 MFMA, %s; lattice-level smoother %s "
                            "degree %d), fp64, tol %.0e"
                            % (ne, nb, "even-odd form (dense 4096^2 Schur inverse)"
                               if tp.get("ref_coarsest") == "eo" else "dense 8192^2 inverse",
                               tp.get("ref_smoother"), tp.get("ref_cycle_post"), args.tol)
                            if args.workload == "config2" else
                            "synthetic %dx%d random U(1) lattice (sigma 0.204, m -0.05), %d x %d plain "
                            "Hutchinson probes/GPU/step, GPU-side adaptive MG setup, fp64, tol %.0e"
                            % (L, L, ne, nb, args.tol) if synthetic else
                            "schwinger128, %d x %d MLMC level-0 difference probes/GPU/step "
                            "(A0^-1 - P0 P1 A2^-1 R1 R0, reference hierarchy 32768/8192/2048/512, "
                            "level skipping), fp64, tol %.0e" % (ne, nb, args.tol),
                "probes_per_step_per_gpu": ne * nb,
                "streams_per_gpu": ne,
                "solver": mg.solver_info if mg.solver_info else
                {"levels": levels, "hierarchy": "reference (multigrid.py:100-345)",
                 "smoother": tp.get("ref_smoother", "mr"), "nu_post": tp.get("ref_cycle_post", 4),
                 "restart": tp.get("solver_restart", 24)},
                "outer_iterations_max": max(iters_seen) if iters_seen else None,
                "trace_estimate": [float(np.real(mean + tr1)), float(np.imag(mean + tr1))],
                "std_dev": std,
                "setup_s": t_setup,
            },
            # dominant kernel of the step (by device time)
            "roofline": dominant,
            # the north star's named target: the batched Dirac stencil against the HBM roofline
            "stencil_roofline": dict(stencil or {}, back_to_back_ms=dirac_ms,
                                     back_to_back_GBs=bytes0 / (dirac_ms * 1e-3) / 1e9,
                                     back_to_back_frac=bytes0 / (dirac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
            "kernel_rooflines": rooflines,
            "step_breakdown_ms": dict(buckets, kernel_launches=launches),
        }
        if not args.no_large_stencil and not synthetic:
            out["stencil_roofline_1024"] = large_stencil_point()
        if world == 1 and not args.no_cpu_baseline and args.workload == "hutchinson":  # 128^2 only
            out["cpu_baseline"] = cpu_baseline(A, tp, mg, args.cpu_probes, args.cpu_workers)
        if world == 1 and args.workload == "hutchinson" and not args.no_other_configs:
            out["other_configs"] = other_configs(args, mg, n, device_index)
        line_out = json.dumps(out)
    if td.is_initialized():
        comm.barrier()
        td.destroy_process_group()
    return line_out


def other_configs(args, mg, n, device_index):
    """Compact records of the BASELINE configurations the headline is not quoted on, measured in the SAME
    driver-run process (one engine, one stream each): so that they are driver-observed numbers,
    not builder files.  Each carries its throughput, iteration count, launch count and dominant kernel."""
    import contextlib
    import io
    import tempfile
    from deflatedmlmc_schwinger_amd.engine import MODE_HUTCHINSON, MODE_MLMC_SKIP
    out = {}
    t_all = time.time()

    def guarded(key, fn):
        t0 = time.time()
        try:
            out[key] = fn()
        except Exception as exc:            # a secondary record must never cost the headline line
            out[key] = {"error": repr(exc)}
        if isinstance(out[key], dict):
            out[key]["wall_s"] = round(time.time() - t0, 2)

    # the headline workload again at the reference's own stopping point (stop_factor = 1), one engine, with
    # its kernel breakdown (the driver-timed K-step figure is the top-level value_reference_stopping_point)
    if args.stop_factor != 1.0:
        guarded("reference_stopping_point_stop_factor_1.0", lambda: secondary(
            "headline workload with engine option stop_factor = 1 (residual below tol: where the reference's "
            "fgmres stops)", mg, MODE_HUTCHINSON, args.nb, n, args.tol, 4, 1, {"stop_factor": 1.0}))
    # config 3: MLMC level-0 difference probes (level skipping, reference hierarchy 32768/8192/2048/512)
    guarded("config3_mlmc_level0_difference", lambda: secondary(
        "schwinger128 MLMC level-0 difference probes A0^-1 - P0 P1 A2^-1 R1 R0 (reference hierarchy "
        "32768/8192/2048/512, level skipping)", mg, MODE_MLMC_SKIP, args.nb, n, args.tol, 4, 1))

    def config2():
        mg2, A2, tp2, _, n2, ts = build_problem("config2", args, device_index, 1)
        try:
            r = secondary("BASELINE config 2 as written: plain Hutchinson (k=0), 2-level multigrid 32768 -> "
                          "8192 (reference aggregation), coarse level solved exactly on fp64 MFMA (%s)"
                          % ("even-odd form: dense 4096^2 inverse of its Schur complement"
                             if tp2.get("ref_coarsest") == "eo" else "dense 8192^2 inverse"),
                          mg2, MODE_HUTCHINSON, args.nb, n2, args.tol, 2, 1)
            r["setup_s"] = ts
            r["smoother"] = {"kind": tp2.get("ref_smoother"), "degree": tp2.get("ref_cycle_post"),
                             "restart": tp2.get("solver_restart"), "coarsest": tp2.get("ref_coarsest")}
            return r
        finally:
            for e_ in mg2.engines:
                e_.close()
    guarded("config2_as_written", config2)

    def config5(levels3):
        a5 = argparse.Namespace(**vars(args))
        a5.lattice = 1024
        a5.cfg = ""
        a5.synthetic_levels = 3 if levels3 else 0
        mg5, _, _, _, n5, ts = build_problem("synthetic", a5, device_index, 1)
        try:
            r = secondary("BASELINE config 5's lattice: synthetic 1024^2 random U(1) gauge field (sigma 0.204, "
                          "m -0.05), 128 plain Hutchinson probes per batch, "
                          + ("THREE-level hierarchy 2097152 -> 262144 -> 4096 (config 5 as written)" if levels3
                             else "five-level hierarchy") + " built on the GPU",
                          mg5, MODE_HUTCHINSON, 128, n5, args.tol, 2, 1)
            r["setup_s"] = ts
            r["setup_log"] = (mg5.solver_info or {}).get("setup_log")
            r["cycle"] = ((mg5.solver_info or {}).get("cfg") or {}).get("cycle")
            return r
        finally:
            for e_ in mg5.engines:
                e_.close()
    guarded("config5_synthetic_1024", lambda: config5(False))
    guarded("config5_three_level", lambda: config5(True))

    # the drop-in flows as main.py:20-22 runs them (gateway.G202 / G102: setup, deflation vectors, rough
    # trace, probe loops with the sequential stopping rule), wall clock and probe-loop throughput; the ARPACK
    # products are shared between the two through the on-disk cache (SW_CACHE_DIR), the first run fills it
    def flow(name):
        from deflatedmlmc_schwinger_amd import gateway
        rep = os.path.join(cache, name + ".jsonl")
        os.environ["SW_REPORT_PATH"] = rep
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            res = getattr(gateway, name)()
        dt = time.time() - t0
        rec = json.loads(open(rep).read().splitlines()[-1])
        keep = {k: rec[k] for k in ("trace", "elapsed_s", "nr_ests", "std_dev", "probes_solved", "probe_loop_s",
                                    "probe_loop_solved_per_s", "probe_loop_used_per_s", "levels") if k in rec}
        keep["wall_s_with_setup"] = dt
        keep["exact_trace_gateway_py_104"] = [gateway.EXACT_TRACE_SCHWINGER128.real,
                                              gateway.EXACT_TRACE_SCHWINGER128.imag]
        del res
        return keep
    cache = tempfile.mkdtemp(prefix="sw_bench_cache_")
    old_env = {k: os.environ.get(k) for k in ("SW_CACHE_DIR", "SW_REPORT_PATH")}
    os.environ["SW_CACHE_DIR"] = cache
    try:
        guarded("dropin_G202_mlmc", lambda: flow("G202"))
        guarded("dropin_G102_hutchinson", lambda: flow("G102"))
    finally:
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        import shutil
        shutil.rmtree(cache, ignore_errors=True)
    out["total_wall_s"] = round(time.time() - t_all, 1)
    return out


def large_stencil_point(L=1024, nb=64, reps=20):
    """HBM roofline point of the stencil beyond the 256 MB Infinity Cache (BASELINE config 5's
    lattice): synthetic random U(1) links on L^2 sites, nb probes, back-to-back launches."""
    from deflatedmlmc_schwinger_amd import matrix as swm
    from deflatedmlmc_schwinger_amd.engine import Engine
    U1, U2 = swm.synthetic_links(L, 0.45, 2024)
    import torch
    eng = Engine(torch.cuda.current_device())
    eng.hier_begin(0, 1)
    eng.set_lattice(0, L, -0.05, U1, U2)
    eng.hier_end(0)
    ms = eng.bench_dirac(0, 0, nb, reps)
    nbp = ((nb + 63) // 64) * 64
    work = L * L * (64.0 * nbp + 32.0)
    eng.close()
    ach = work / (ms * 1e-3) / 1e9
    return {"kernel": "k_stencil<0>", "workload": "synthetic %dx%d, %d probes, back-to-back" % (L, L, nb),
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "work_per_launch": work, "avg_launch_ms": ms}


def _oracle_probe_loop(A, tp, testvectors, Ux, nprobes, seed):
    """`nprobes` deflated Hutchinson probes through the oracle's reference-faithful path (SciPy
    CSR + lgmres(maxiter=2) smoother + flexible GMRES, tol 1e-12); returns the loop's seconds.
    The probe draw (np.random.randint, utils.py:213-216) is inside the timed loop, as the probe
    generation is inside the GPU's timed region."""
    from oracle import ref_path as rp
    omg = rp.OracleMG(A)
    omg.setup(tp['dof'], tp['aggrs'], tp['max_nr_levels'], tp['accuracy_mg_eigvs'], tp,
              testvectors=testvectors)
    PT = omg.ml.levels[0].Pperm.transpose()

    def solve(b):
        omg.level_nr = 0
        omg.solve(A, b, tp['function_params']['tol'])
        return omg.x

    def loop():
        np.random.seed(seed)
        t0 = time.perf_counter()
        for _ in range(nprobes):
            rp.hutch_probe(rp.rademacher(A.shape[0]), solve, Ux, PT)
        return time.perf_counter() - t0
    return loop


def _cpu_worker(A, tp, testvectors, Ux, nprobes, seed, barrier, queue):
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        loop = _oracle_probe_loop(A, tp, testvectors, Ux, nprobes, seed)
        barrier.wait(timeout=600)
        queue.put(loop())
    except Exception as exc:      # the parent reports it
        queue.put("error: %r" % (exc,))


def cpu_baseline(A, tp, mg, nprobes, nworkers):
    """The oracle's reference-faithful path on this host: the headline is
    ONE core with OMP_NUM_THREADS=1 as main.py:20 forces; the second line uses the host's cores the
    only way that helps this workload (threads inside SciPy/BLAS slow it down):
    independent probes in `nworkers` single-threaded processes."""
    from oracle import ref_path as rp
    lev0_g3 = mg.ml.levels[0].g3
    Ux, _, _, _ = rp.deflation_hutchinson(A, lev0_g3, mg.ml.levels[0].Pperm, tp['nr_deflat_vctrs'],
                                          tp['defl_eigvs_tol_Hutch'], True)
    tp_plain = {k: v for k, v in tp.items() if k not in ("mg_testvectors", "solver_testvectors")}
    dt = _oracle_probe_loop(A, tp_plain, mg.testvectors, Ux, nprobes, 123456)()
    out = {"value": nprobes / dt, "unit": "probe-samples/s", "cores": 1, "kind": "port",
           "sample": "%d deflated Hutchinson probes of the same workload (probe draw included, setup "
                     "excluded), oracle/ref_path.py restatement of the reference path, 1 thread of %d "
                     "host cores" % (nprobes, os.cpu_count() or 0),
           "seconds": dt}
    if nworkers < 0:
        nworkers = min(os.cpu_count() or 1, 16)
    if nworkers > 1:
        import multiprocessing as mp
        ctx = mp.get_context("spawn")          # fresh interpreters: NumPy/SciPy only, no GPU
        per = 2
        barrier = ctx.Barrier(nworkers)
        queue = ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker,
                             args=(A, tp_plain, mg.testvectors, Ux, per, 1000 + w, barrier, queue))
                 for w in range(nworkers)]
        for pr in procs:
            pr.start()
        times = []
        try:
            for _ in procs:
                r = queue.get(timeout=900)
                if isinstance(r, str):
                    raise RuntimeError(r)
                times.append(r)
            out["all_cores"] = {"value": nworkers * per / max(times), "unit": "probe-samples/s",
                                "cores": nworkers, "kind": "port",
                                "sample": "%d single-threaded processes x %d probes each (independent "
                                          "probe streams), same path" % (nworkers, per),
                                "seconds": max(times)}
        except Exception as exc:
            out["all_cores"] = {"error": repr(exc)}
        finally:
            for pr in procs:
                pr.join(timeout=30)
                if pr.is_alive():
                    pr.terminate()
    return out


if __name__ == "__main__":
    main()
