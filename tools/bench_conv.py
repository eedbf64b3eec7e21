"""Runs that stop by the reference's convergence criterion (decoders.py:74-82): the work-queue kernel against one ladder per
lane (QECMC_QUEUE_GRID huge = every ladder gets its own lane from the start, i.e. the round-1 behaviour with early exit per
workgroup).  Toric L=5 p=0.10 Nc=5, default criterion (SEQ=2, TOPS=10, eps=0.1), horizon 2^17 steps."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, "mcmc-qec-toric-rl_amd")
sys.path.insert(0, ".")
import qecmc
import bench

L, p, Nc, N, H = (int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (5, 0.10, 5, 1 << 18, 1 << 16)
init = bench.synth_batch(N, L, p, 7)
out = {}
for name, grid in (("queue", None), ("one_ladder_per_lane", str(1 << 30))):
    if grid: os.environ["QECMC_QUEUE_GRID"] = grid
    else: os.environ.pop("QECMC_QUEUE_GRID", None)
    for rep in range(1 if L > 5 else 2):
        t0 = time.time()
        r = qecmc.pteq_batch(init, p, Nc=Nc, steps=H, iters=10, tops_burn=2, seed=3, conv_criteria="error_based", return_stats=True)
        dt = time.time() - t0
    steps = r["steps_done"].astype(np.float64)
    out[name] = dict(kernel_ms=r["stats"]["kernel_ms"], wall_s=dt, converged_frac=float(r["converged"].mean()), mean_steps=float(steps.mean()),
                     median_steps=float(np.median(steps)), p99_steps=float(np.percentile(steps, 99)), max_steps=float(steps.max()),
                     useful_proposals_per_s=float(steps.sum() * Nc * 10 / (r["stats"]["kernel_ms"] * 1e-3)),
                     checksum=int(r["counts"].astype(np.uint64).sum()))
    print(name, json.dumps(out[name]), flush=True)
out["speedup"] = out["one_ladder_per_lane"]["kernel_ms"] / out["queue"]["kernel_ms"]
out["workload"] = "toric L=%d p=%g Nc=%d, %d syndromes, error_based criterion (SEQ=2, TOPS=10, eps=0.1), horizon %d ladder steps" % (L, p, Nc, N, H)
json.dump(out, open("gpurun_out/r02_conv_queue_L%d.json" % L, "w"), indent=1)
print("speedup", out["speedup"])
