"""Syndrome shards across the GPUs of one node.

The reference parallelises over syndromes only at job level (SLURM array tasks,
generate_data.py:274-276, merged offline by concat_data.py).  Here one process
drives one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI); the batch
is cut into contiguous shards, every rank runs the ladder kernel on its shard
with Philox keyed by the GLOBAL syndrome index (so the answer does not depend on
the number of GPUs), and the only exchange is one gather of the per-class
counts -- 4*ncls + 8 bytes per syndrome, latency-bound on a fully connected
xGMI node, no collective on the sampling path itself.
"""
import numpy as np


def shard_bounds(n_total, world_size, rank):
    """Contiguous shard [lo, hi) of `rank`; sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pteq_batch_sharded(init, p, compute=None, group=None, dst=0, **kw):
    """PTEQ on a batch sharded over the ranks of `group`.

    init: the FULL uint8[N,2,L,L] batch (every rank passes the same array; only its shard is used).
    compute(init_shard, p, first_syndrome=..., **kw) -> dict(counts, samples, tops0): defaults to
    qecmc.pteq_batch (the GPU path); tests inject a stand-in to exercise the exchange on CPU.
    Returns the gathered dict on rank `dst` (arrays in global syndrome order) and None elsewhere.
    """
    import torch
    import torch.distributed as dist
    if compute is None:
        from .decoders import pteq_batch as compute
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = int(np.asarray(init).shape[0])
    lo, hi = shard_bounds(n_total, world, rank)
    first = int(kw.pop("first_syndrome", 0))
    if dist.get_backend(group) == "nccl":
        kw.setdefault("device", torch.cuda.current_device())
    res = compute(np.asarray(init)[lo:hi], p, first_syndrome=first + lo, **kw)
    ncls = res["counts"].shape[1]
    # one packed record per syndrome: ncls class counts + samples + tops0 (uint32 -> int64-safe int32 view)
    rec = np.concatenate([res["counts"].astype(np.uint32), res["samples"].astype(np.uint32)[:, None],
                          res["tops0"].astype(np.uint32)[:, None]], axis=1)
    max_rows = (n_total + world - 1) // world          # equal-size buffers for the collective
    buf = np.zeros((max_rows, ncls + 2), dtype=np.uint32)
    buf[:hi - lo] = rec
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.from_numpy(buf.view(np.int32)).to(dev)
    gathered = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.zeros((n_total, ncls + 2), dtype=np.uint32)
    for r in range(world):
        rlo, rhi = shard_bounds(n_total, world, r)
        out[rlo:rhi] = gathered[r].cpu().numpy().view(np.uint32)[:rhi - rlo]
    return dict(counts=out[:, :ncls].copy(), samples=out[:, ncls].copy(), tops0=out[:, ncls + 1].copy())
