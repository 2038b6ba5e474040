"""CPU: the N > 1 probe-sharding path with world_size 2 over gloo.  The GPU evaluator is
replaced by an exact host evaluator (oracle LU on the 16^2 lattice); what is under test is the
product's sharding / gathering / replay code (dist.py + stoch_trace.run_probe_loop)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_eval():
    sys.path.insert(0, ROOT)
    from deflatedmlmc_schwinger_amd import gateway, matrix
    from oracle import ref_path as rp
    p = gateway.set_params('schwinger16')
    A = matrix.loadMatrix(p['matrix'], p['matrix_params'])
    lu = rp.LUSolver(A)

    def evaluate(probes):
        e = np.array([rp.hutch_probe(x.astype(np.complex128), lu, None, None) for x in probes])
        return e, np.full(len(e), 3), np.zeros(len(e), dtype=np.int64)
    return A.shape[0], evaluate


def _worker(rank, world, port, tol, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from deflatedmlmc_schwinger_amd import dist, stoch_trace
    n, evaluate = _make_eval()
    comm = dist.TorchComm()
    assert comm.my_slice(10) == ((0, 5) if rank == 0 else (5, 10))
    np.random.seed(123456)
    out = stoch_trace.run_probe_loop(evaluate, n, tol, 100000, batch, comm=comm)
    stats = comm.allreduce_stats(dist.local_stats(out["ests"][rank::world]))
    q.put((rank, out["index"], complex(out["avg"]), float(out["dev"]), out["ests"].tolist(),
           int(out["iters_fine"].sum()), stats.tolist(), int(np.random.randint(1 << 30))))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_probe_loop_equals_single_process():
    sys.path.insert(0, ROOT)
    from deflatedmlmc_schwinger_amd import dist, stoch_trace
    n, evaluate = _make_eval()
    tol = 6.0
    np.random.seed(123456)
    ref = stoch_trace.run_probe_loop(evaluate, n, tol, 100000, 16, comm=dist.Comm())
    ref_next = int(np.random.randint(1 << 30))
    assert ref["index"] >= 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tol, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, idx, avg, dev, ests, its, stats, nxt in results:
        assert idx == ref["index"]
        assert avg == complex(ref["avg"]) and dev == float(ref["dev"])
        assert np.array_equal(np.array(ests), ref["ests"])
        assert its == int(ref["iters_fine"].sum())
        assert nxt == ref_next                       # global stream left where 1 process leaves it
        mean, std = dist.mean_and_population_std(stats)
        assert abs(mean - ref["avg"]) < 1e-9 * abs(ref["avg"])
        assert abs(std - ref["dev"]) < 1e-9 * ref["dev"]


def _sha(arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    return h.hexdigest()


def _worker_setup_and_mlmc(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from deflatedmlmc_schwinger_amd import dist, gateway, matrix, multigrid, stoch_trace, utils
    from oracle import ref_path as rp
    comm = dist.TorchComm()
    # 1. compute_on_root: every rank ends with rank 0's bytes even when the local results differ
    mine = np.random.default_rng(100 + rank).standard_normal((7, 3)) + 1j * rank
    got = comm.compute_on_root(lambda: (mine, {"k": mine[:2].copy()}))
    sha_bcast = _sha([got[0], got[1]["k"]])
    # 2. the collective hierarchy build: test vectors from rank 0, operators identical
    p = gateway.set_params('schwinger16')
    p['function_tol'] = 1e-12
    A = matrix.loadMatrix(p['matrix'], p['matrix_params'])
    tp = utils.trace_params_from_params(p, "mlmc")
    ml, cinv, used = multigrid.collective_reference_hierarchy(A, tp['dof'], tp['aggrs'],
                                                              tp['max_nr_levels'], 'high', tp,
                                                              comm=comm)
    ops = list(used) + [np.asarray(cinv)]
    for lev in ml.levels[:-1]:
        for M in (lev.P.tocsr(), lev.A.tocsr()):
            ops += [M.indptr, M.indices, M.data]
    sha_hier = _sha(ops)
    # 3. the MLMC branch of the probe loop (level-0 difference, exact host evaluator)
    cinv_a = np.asarray(cinv)
    lus = {l: rp.LUSolver(ml.levels[l].A) for l in range(2)}

    def evaluate(probes):
        e = np.array([rp.mlmc_probe(x.astype(np.complex128), 0, ml.levels, False,
                                    lambda l, b: lus[l](b), cinv_a, False) for x in probes])
        return e, np.full(len(e), 5), np.full(len(e), 2)
    np.random.seed(4242)
    out = stoch_trace.run_probe_loop(evaluate, A.shape[0], 0.35, 100000, 8, comm=comm)
    q.put((rank, sha_bcast, sha_hier, out["index"], complex(out["avg"]), float(out["dev"]),
           out["ests"].tolist(), int(out["iters_coarse"].sum()), int(np.random.randint(1 << 30))))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_share_setup_operands_and_replay_mlmc_branch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_setup_and_mlmc, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = results
    assert r0[1] == r1[1]                 # broadcast object: byte-identical
    assert r0[2] == r1[2]                 # test vectors, P, A, coarsest inverse: byte-identical
    assert r0[3:] == r1[3:]               # MLMC loop: same stop index, stats, values, stream position
    assert r0[3] >= 5 and r0[7] == 2 * (r0[3] + 1)


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    from deflatedmlmc_schwinger_amd import dist, stoch_trace
    comm = dist.TorchComm()
    # (a) contiguous slices of ragged rounds tile the round
    cover = []
    for count in (8, 13, 64, 257):
        lo, hi = comm.my_slice(count)
        cover.append((count, lo, hi))
    # (b) the packed all-gather of per-probe results, ragged round (some ranks hold fewer probes)
    count = 13
    lo, hi = comm.my_slice(count)
    idx = np.arange(lo, hi)
    e, f, c = comm.allgather_probe_results(idx * (1.0 - 2.0j), 100 + idx, 7 * idx, count)
    ok_gather = (np.array_equal(e, np.arange(count) * (1.0 - 2.0j)) and
                 np.array_equal(f, 100 + np.arange(count)) and np.array_equal(c, 7 * np.arange(count)))
    # (c) bench.py's block map: the (rank, stream) blocks of a round tile the probe stream
    mine = [bench.first_probe(s, world, rank, 3, e_, 256) for s in range(2) for e_ in range(3)]
    # (d) the probe loop itself with a synthetic evaluator whose estimate encodes the probe's entries
    n = 96

    def evaluate(probes):
        # integer arithmetic: exactly the same value for a probe whatever batch it arrives in
        p = np.asarray(probes, dtype=np.int64)
        w = (np.arange(n) % 7) - 3
        return (p * w).sum(axis=1) * (3.0 + 1.0j) + 10.0, np.full(len(p), 2), np.zeros(len(p), dtype=np.int64)
    np.random.seed(2718)
    out = stoch_trace.run_probe_loop(evaluate, n, 0.45, 137, 5, comm=comm)
    q.put((rank, cover, bool(ok_gather), mine, out["index"], complex(out["avg"]), float(out["dev"]),
           out["ests"].tolist(), int(np.random.randint(1 << 30))))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_rank_partitioning_gather_and_probe_loop():
    """world size 8 (what the driver's scaling run uses) over gloo on the CPU: slice partitioning of
    ragged rounds, the packed per-probe all-gather, bench.py's (rank, stream) block map, and the probe
    loop against the single-process loop -- same stop index, statistics, values and stream position."""
    sys.path.insert(0, ROOT)
    from deflatedmlmc_schwinger_amd import dist, stoch_trace
    n = 96

    def evaluate(probes):
        # integer arithmetic: exactly the same value for a probe whatever batch it arrives in
        p = np.asarray(probes, dtype=np.int64)
        w = (np.arange(n) % 7) - 3
        return (p * w).sum(axis=1) * (3.0 + 1.0j) + 10.0, np.full(len(p), 2), np.zeros(len(p), dtype=np.int64)
    np.random.seed(2718)
    ref = stoch_trace.run_probe_loop(evaluate, n, 0.45, 137, 5, comm=dist.Comm())
    ref_next = int(np.random.randint(1 << 30))
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for count in (8, 13, 64, 257):
        spans = sorted((lo, hi) for r in results for (c, lo, hi) in r[1] if c == count)
        assert spans[0][0] == 0 and spans[-1][1] == count
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))          # contiguous, no overlap
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert all(r[2] for r in results)
    blocks = sorted(b for r in results for b in r[3])
    assert blocks == [256 * i for i in range(2 * world * 3)]                 # two rounds, no gap, no overlap
    for r in results:
        assert r[4] == ref["index"] and r[5] == complex(ref["avg"]) and r[6] == float(ref["dev"])
        assert np.array_equal(np.array(r[7]), ref["ests"])
        assert r[8] == ref_next
