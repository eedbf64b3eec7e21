"""One-GPU rehearsal of the library's multi-GPU entry: qecmc.sharding.launch starts the ranks (here 1: the box has one GPU)
from a parent that has not touched the GPU, each rank runs pteq_batch_sharded (PteqShard: kernel + RCCL gather from device
memory), rank 0 returns the gathered result, and a second launch computes the unsharded call for comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))


def make(n=300, L=5, p=0.1):
    rng = np.random.default_rng(1)
    m = np.zeros((n, 2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


def sharded(rank, world):
    sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
    from qecmc.sharding import pteq_batch_sharded
    return pteq_batch_sharded(make(), 0.1, Nc=5, steps=200, iters=10, tops_burn=1, seed=3, first_syndrome=7)


def plain(rank, world):
    sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
    import qecmc
    r = qecmc.pteq_batch(make(), 0.1, Nc=5, steps=200, iters=10, tops_burn=1, seed=3, first_syndrome=7)
    return {k: r[k] for k in ("counts", "samples", "tops0")}


if __name__ == "__main__":
    from qecmc.sharding import launch
    a = launch(1, sharded, backend="nccl", timeout=300)
    b = launch(1, plain, backend="nccl", timeout=300)
    ok = all(np.array_equal(a[k], b[k]) for k in ("counts", "samples", "tops0"))
    print("sharding.launch + PteqShard + RCCL gather (world 1) equals pteq_batch:", ok, a["counts"].shape)
    sys.exit(0 if ok else 1)
