"""Syndrome shards across the GPUs of one node.

The reference parallelises over syndromes only at job level (SLURM array tasks,
generate_data.py:274-276, merged offline by concat_data.py).  Here one process
drives one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI); the batch
is cut into contiguous shards, every rank runs the ladder kernel on its shard
with Philox keyed by the GLOBAL syndrome index (so the answer does not depend on
the number of GPUs), and the only exchange is one gather of the per-class
counts -- 4*ncls + 8 bytes per syndrome, latency-bound on a fully connected
xGMI node, no collective on the sampling path itself.

    launch(n, fn, ...)            start n ranks on this node (the array job of generate_data.py:274-276 -> one node)
    PteqShard                     one rank's shard resident in HBM: .launch() (kernel) and .gather() (the exchange)
    pteq_batch_sharded(init, p)   host-array convenience call over PteqShard
"""
import os
import socket

import numpy as np


def shard_bounds(n_total, world_size, rank, align=1):
    """Contiguous shard [lo, hi) of `rank`; sizes differ by at most one unit of `align` syndromes (align = 64 for scan = "wave", whose
    wavefronts share their generator picks: a shard must begin on a multiple of 64 of the global index; the last shard takes the ragged end)."""
    n_total, align = int(n_total), int(align)
    units = (n_total + align - 1) // align
    base, extra = divmod(units, int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return min(lo * align, n_total), min(hi * align, n_total)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, backend, fn, args, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    import torch
    import torch.distributed as dist
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(rank)
        kw["device_id"] = torch.device("cuda", rank)
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    try:
        res = fn(rank, world, *args)
        if rank == 0:
            q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def launch(n, fn, args=(), backend="nccl", timeout=3600):
    """Run fn(rank, world, *args) on n fresh processes of this node, one per GPU (rank r drives cuda:r), inside an initialised
    process group (backend "nccl" = RCCL; "gloo" for CPU rehearsals), and return rank 0's return value.

    The ranks are started with the `spawn` method BEFORE this process has touched the GPU: a process that has initialised HIP
    must neither fork workers (HIP is not fork-safe) nor be replaced by another program.  `fn` and `args` must be picklable
    (a module-level function).  Any rank failing raises RuntimeError with its exit code."""
    import torch
    import torch.multiprocessing as mp
    if backend == "nccl" and torch.cuda.is_initialized():
        raise RuntimeError("sharding.launch must be called before this process initialises the GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, n, port, backend, fn, args, q)) for r in range(n)]
    for p in procs:
        p.start()
    # wait for rank 0's result while watching the children: a rank that dies (GPU fault, import error, an exception in fn) leaves
    # the others in a barrier, so on the first non-zero exit code the remaining ranks are ended and the failure raised at once
    import queue as _queue
    import time as _time
    res, got = None, False
    deadline = _time.monotonic() + timeout

    def failed():
        return [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode not in (None, 0)]
    while not got and not failed() and _time.monotonic() < deadline:
        try:
            res = q.get(timeout=0.5)
            got = True
        except _queue.Empty:
            if all(p.exitcode is not None for p in procs):
                break                                        # everyone has left and rank 0 put nothing
    # rank 0 has delivered: the others only have the closing barrier left
    end = _time.monotonic() + (60 if got else 0)
    while got and not failed() and any(p.exitcode is None for p in procs) and _time.monotonic() < end:
        _time.sleep(0.05)
    bad = failed()
    for p in procs:
        if p.is_alive():
            p.kill()
    for p in procs:
        p.join(5)
    if bad or not got:
        raise RuntimeError(f"sharding.launch: ranks failed (rank, exit code): {bad or 'no result from rank 0'}")
    return res


class PteqShard:
    """This rank's shard of a PTEQ batch, resident in HBM: plan, input and one packed record buffer
    (ncls class counts, samples, tops0 per syndrome).  `launch()` enqueues the ladder kernel on the current stream,
    `gather()` is the path's one exchange (per-class counts -> rank `dst`).  bench.py times exactly these two calls."""

    def __init__(self, init_shard, p, first_syndrome, n_total=None, group=None, dst=0, device=None, **params):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib as L_
        self._C, self._L, self._torch, self._dist = C, L_, torch, dist
        self.group, self.dst = group, dst
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist_on else 1
        self.rank = dist.get_rank(group) if self.dist_on else 0
        dev = torch.cuda.current_device() if device is None else device
        self.dev = torch.device("cuda", dev)
        a = np.ascontiguousarray(init_shard, dtype=np.uint8)
        self.n = a.shape[0]
        code = params.get("code", L_.TORIC)
        self.ncls = 16 if code == L_.TORIC else 4
        size = a.shape[-1]
        params.setdefault("p_logical", 0.5)            # decoders.py:52
        params.setdefault("steps", 1000)
        params["Nc"] = params.get("Nc") or size        # decoders.py:30
        self.pr = L_.make_params(L=size, p=float(p), device=dev, first_syndrome=0, **params)
        self.first = int(first_syndrome)
        self.plan = C.c_void_p()
        L_.check(L_.lib().qecmc_plan_create(self.pr, C.byref(self.plan)))
        self.d_init = torch.from_numpy(a.reshape(self.n, int(np.prod(a.shape[1:])))).to(self.dev)
        self.n_total = int(n_total if n_total is not None else self.n * self.world)
        if isinstance(params.get("scan"), str):
            params["scan"] = L_.SCANS[params["scan"]]
        self.align = 64 if params.get("scan") == L_.SCAN_WAVE else 1
        self.max_rows = max(shard_bounds(self.n_total, self.world, r, self.align)[1] - shard_bounds(self.n_total, self.world, r, self.align)[0]
                            for r in range(self.world))                                # equal-size buffers for the collective
        self.rec = torch.zeros(self.max_rows * (self.ncls + 2), dtype=torch.int32, device=self.dev)
        self.gathered = ([torch.empty_like(self.rec) for _ in range(self.world)]
                         if (self.dist_on and self.rank == dst) else None)
        # runs that stop by the criterion (decoders.py:74-82): the log workspace, and where every ladder stopped
        self.ws = self.steps_done = self.converged = None
        self.ws_bytes = 0
        if params.get("conv_mode", L_.CONV_NONE) != L_.CONV_NONE:
            need = C.c_uint64()
            L_.check(L_.lib().qecmc_plan_workspace_bytes(self.plan, self.n, 0, C.byref(need)))
            self.ws_bytes = int(need.value)
            self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=self.dev)
            self.steps_done = torch.zeros(self.n, dtype=torch.int32, device=self.dev)
            self.converged = torch.zeros(self.n, dtype=torch.uint8, device=self.dev)

    def views(self, rec=None, n=None):
        rec = self.rec if rec is None else rec
        n = self.n if n is None else n
        k = n * self.ncls
        return rec[:k].view(n, self.ncls), rec[k:k + n], rec[k + n:k + 2 * n]

    def launch(self, stream=None):
        C, L_, torch = self._C, self._L, self._torch
        stream = torch.cuda.current_stream(self.dev) if stream is None else stream
        counts, samples, tops0 = self.views()
        if self.n:
            L_.check(L_.lib().qecmc_pteq_launch_dev(self.plan, self.d_init.data_ptr(), self.n, self.first, counts.data_ptr(),
                                                    samples.data_ptr(), tops0.data_ptr(),
                                                    None if self.steps_done is None else self.steps_done.data_ptr(),
                                                    None if self.converged is None else self.converged.data_ptr(), None,
                                                    None if self.ws is None else self.ws.data_ptr(), self.ws_bytes,
                                                    C.c_void_p(stream.cuda_stream)))

    def gather(self):
        if self.dist_on:
            self._dist.gather(self.rec, self.gathered, dst=self.dst, group=self.group)

    def result(self):
        """On rank `dst`: dict(counts, samples, tops0) in global syndrome order (after gather()); None elsewhere."""
        if self.rank != self.dst:
            return None
        recs = self.gathered if self.dist_on else [self.rec]
        out = [[], [], []]
        for r, rec in enumerate(recs):
            lo, hi = shard_bounds(self.n_total, self.world, r, self.align) if self.dist_on else (0, self.n)
            for o, v in zip(out, self.views(rec, hi - lo)):
                o.append(v.cpu().numpy().view(np.uint32))
        return dict(counts=np.concatenate(out[0]), samples=np.concatenate(out[1]), tops0=np.concatenate(out[2]))

    def close(self):
        if self.plan:
            self._L.lib().qecmc_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pteq_batch_sharded(init, p, compute=None, group=None, dst=0, **kw):
    """PTEQ on a batch sharded over the ranks of `group`.

    init: the FULL uint8[N,2,L,L] batch (every rank passes the same array; only its shard is used).
    Default (compute=None): the GPU path -- PteqShard: shard to HBM, ladder kernel, one RCCL gather of the packed records
    straight from device memory.  compute(init_shard, p, first_syndrome=..., **kw) -> dict(counts, samples, tops0) is the
    tests' hook to exercise bounds and exchange on CPU with a stand-in.
    Returns the gathered dict on rank `dst` (arrays in global syndrome order) and None elsewhere.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = int(np.asarray(init).shape[0])
    align = 64 if kw.get("scan") in ("wave", 3) else 1
    lo, hi = shard_bounds(n_total, world, rank, align)
    first = int(kw.pop("first_syndrome", 0))
    if compute is None:
        kw.pop("device", None)
        sh = PteqShard(np.asarray(init)[lo:hi], p, first + lo, n_total=n_total, group=group, dst=dst, **kw)
        sh.launch()
        sh.gather()
        torch.cuda.synchronize()
        out = sh.result()
        sh.close()
        return out
    res = compute(np.asarray(init)[lo:hi], p, first_syndrome=first + lo, **kw)
    ncls = res["counts"].shape[1]
    # one packed record per syndrome: ncls class counts + samples + tops0 (uint32 -> int64-safe int32 view)
    rec = np.concatenate([res["counts"].astype(np.uint32), res["samples"].astype(np.uint32)[:, None],
                          res["tops0"].astype(np.uint32)[:, None]], axis=1)
    max_rows = max(shard_bounds(n_total, world, r, align)[1] - shard_bounds(n_total, world, r, align)[0] for r in range(world))   # equal-size buffers for the collective
    buf = np.zeros((max_rows, ncls + 2), dtype=np.uint32)
    buf[:hi - lo] = rec
    t = torch.from_numpy(buf.view(np.int32))
    if dist.get_backend(group) == "nccl":
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.zeros((n_total, ncls + 2), dtype=np.uint32)
    for r in range(world):
        rlo, rhi = shard_bounds(n_total, world, r, align)
        out[rlo:rhi] = gathered[r].cpu().numpy().view(np.uint32)[:rhi - rlo]
    return dict(counts=out[:, :ncls].copy(), samples=out[:, ncls].copy(), tops0=out[:, ncls + 1].copy())
