"""Probe sharding across GPUs: one process per GPU, replicated operands,
contiguous probe ranges per rank, ONE small collective per round.

Every rank draws the SAME seeded probe stream and keeps only its contiguous slice, so the
stream position of probe k is identical to the single-GPU run.  After a round the ranks
all-gather the 16-byte per-probe estimates (so the sequential stopping rule of
stoch_trace.py:137-154 can be replayed identically everywhere) and all-reduce the four
running statistics {sum Re e, sum Im e, sum |e|^2, count}.  ``torch.distributed`` supplies
the transport: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in CPU tests.
"""
import numpy as np


class Comm:
    """Trivial single-process communicator."""
    world = 1
    rank = 0

    def my_slice(self, count):
        return 0, count

    def allgather(self, arrays, count):
        return arrays

    def allgather_probe_results(self, e, f, c, count):
        return (np.asarray(e, dtype=np.complex128), np.asarray(f, dtype=np.int64),
                np.asarray(c, dtype=np.int64))

    def broadcast_arrays(self, arrays, src=0):
        return arrays

    def compute_on_root(self, fn):
        return fn()

    def allreduce_stats(self, stats):
        return np.asarray(stats, dtype=np.float64)

    def barrier(self):
        pass


class TorchComm(Comm):
    """Communicator over an initialised torch.distributed process group."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as td
        if not td.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch = torch
        self._td = td
        self.world = td.get_world_size()
        self.rank = td.get_rank()
        if device is None:
            device = "cuda" if td.get_backend() == "nccl" else "cpu"
        self.device = device

    def my_slice(self, count):
        lo = (self.rank * count) // self.world
        hi = ((self.rank + 1) * count) // self.world
        return lo, hi

    def allgather(self, arrays, count):
        """arrays: list of 1-D float64/complex128/int arrays holding this rank's slice of a
        round of `count` probes; returns the full-length arrays in probe order."""
        torch, td = self._torch, self._td
        out = []
        sizes = [((r + 1) * count) // self.world - (r * count) // self.world
                 for r in range(self.world)]
        width = max(sizes)
        for a in arrays:
            a = np.asarray(a)
            is_c = np.iscomplexobj(a)
            loc = np.zeros((width, 2), dtype=np.float64)
            if is_c:
                loc[:a.size, 0] = a.real
                loc[:a.size, 1] = a.imag
            else:
                loc[:a.size, 0] = a
            t = torch.from_numpy(loc).to(self.device)
            bufs = [torch.empty_like(t) for _ in range(self.world)]
            td.all_gather(bufs, t)
            parts = []
            for r in range(self.world):
                b = bufs[r].cpu().numpy()[:sizes[r]]
                parts.append(b[:, 0] + 1j * b[:, 1] if is_c else b[:, 0].astype(a.dtype))
            out.append(np.concatenate(parts))
        return out

    def allgather_probe_results(self, e, f, c, count):
        """ONE collective per round: this rank's per-probe (estimate, fine iterations, coarse
        iterations) packed as (width, 4) float64 rows, gathered in probe order."""
        torch, td = self._torch, self._td
        sizes = [((r + 1) * count) // self.world - (r * count) // self.world
                 for r in range(self.world)]
        width = max(sizes)
        e = np.asarray(e, dtype=np.complex128)
        loc = np.zeros((width, 4), dtype=np.float64)
        loc[:e.size, 0] = e.real
        loc[:e.size, 1] = e.imag
        loc[:e.size, 2] = np.asarray(f, dtype=np.float64)      # iteration counts << 2^53: exact
        loc[:e.size, 3] = np.asarray(c, dtype=np.float64)
        t = torch.from_numpy(loc.reshape(-1)).to(self.device)
        out = torch.empty(self.world * width * 4, dtype=torch.float64, device=self.device)
        td.all_gather_into_tensor(out, t)
        full = out.cpu().numpy().reshape(self.world, width, 4)
        parts = [full[r, :sizes[r]] for r in range(self.world)]
        g = np.concatenate(parts, axis=0)
        return (g[:, 0] + 1j * g[:, 1], g[:, 2].astype(np.int64), g[:, 3].astype(np.int64))

    def broadcast_arrays(self, arrays, src=0):
        """Rank `src`'s NumPy arrays on every rank (setup operands: test vectors, eigenpairs),
        byte-identical.  Other ranks pass arrays of the same shape/dtype or None placeholders
        described by (shape, dtype) tuples."""
        torch, td = self._torch, self._td
        out = []
        for a in arrays:
            if isinstance(a, tuple):
                a = np.empty(a[0], dtype=a[1])
            a = np.ascontiguousarray(a)
            t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self.device)
            td.broadcast(t, src=src)
            out.append(t.cpu().numpy().view(a.dtype).reshape(a.shape))
        return out

    def compute_on_root(self, fn, src=0):
        """Setup operands that must be IDENTICAL on every rank (ARPACK test vectors, deflation
        eigenpairs: an iterative eigensolver is not guaranteed bit-reproducible across processes):
        rank `src` computes fn(), every rank returns rank `src`'s result, byte for byte."""
        box = [fn() if self.rank == src else None]
        self._td.broadcast_object_list(box, src=src)
        return box[0]

    def allreduce_stats(self, stats):
        torch, td = self._torch, self._td
        t = torch.tensor(np.asarray(stats, dtype=np.float64), dtype=torch.float64,
                         device=self.device)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t.cpu().numpy()

    def barrier(self):
        if self._td.get_backend() == "nccl":
            # name the device: an NCCL barrier otherwise guesses it from the rank
            self._td.barrier(device_ids=[self._torch.cuda.current_device()])
        else:
            self._td.barrier()


def default_comm():
    """TorchComm when a process group with more than one rank exists, else the trivial one."""
    try:
        import torch.distributed as td
        if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
            return TorchComm()
    except Exception:
        pass
    return Comm()


def local_stats(ests):
    """{sum Re e, sum Im e, sum |e|^2, n} of a set of per-probe estimates."""
    e = np.asarray(ests, dtype=np.complex128)
    return np.array([e.real.sum(), e.imag.sum(), (np.abs(e) ** 2).sum(), float(e.size)])


def mean_and_population_std(stats):
    """Mean and population standard deviation (stoch_trace.py:143-145) from reduced stats:
    var = sum|e|^2/n - |mean|^2."""
    s_re, s_im, s_abs2, n = stats
    mean = complex(s_re, s_im) / n
    var = max(s_abs2 / n - abs(mean) ** 2, 0.0)
    return mean, float(np.sqrt(var))


class EngineComm(TorchComm):
    """TorchComm whose statistics all-reduce goes through the engine's own C-ABI collective
    (`sw_allreduce_stats`: RCCL ncclAllReduce on the engine's stream).  The RCCL unique id travels
    over the existing torch.distributed group (any transport would do)."""

    def __init__(self, engine, device=None):
        super().__init__(device)
        box = [engine.comm_unique_id() if self.rank == 0 else None]
        self._td.broadcast_object_list(box, src=0)
        engine.comm_init(self.world, self.rank, box[0])
        self._engine = engine

    def allreduce_stats(self, stats):
        return self._engine.allreduce_stats(stats)
