"""Throughput of the colour-parallel layout (scan = "colour": one workgroup per ladder, one wave per rung, lanes = the generators of a colour
phase) against the lane-per-chain layout on the same batch: ladders x phases per second and proposals per second.
GPU box:  python tools/colour_throughput.py > gpurun_out/r03_colour_throughput.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
import qecmc as q
from qecmc import _lib as L_


def main():
    rng = np.random.default_rng(3)
    out = {"what": __doc__.split("\n")[0], "rows": []}
    for code, L, p, Nc in (("toric", 9, 0.15, 8), ("toric", 15, 0.18, 8), ("rotated", 21, 0.17, 8)):
        shape = (2, L, L) if code == "toric" else (L, L)
        cid = {"toric": q.TORIC, "rotated": q.ROTATED}[code]
        phases = q.colour_phases(cid, L) if hasattr(q, "colour_phases") else None
        for N in (1, 64, 1024, 4096, 16384):
            init = np.zeros((N,) + shape, dtype=np.uint8)
            err = rng.random(init.shape) < p
            init[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
            row = dict(code=code, L=L, p=p, Nc=Nc, ladders=N)
            for scan, steps in (("colour", 2000), ("random", 2000)):
                if scan == "random" and N < 64:
                    continue
                r = q.pteq_batch(init, p, Nc=Nc, steps=steps, iters=10, tops_burn=2, seed=5, scan=scan, code=cid, return_stats=True)
                r = q.pteq_batch(init, p, Nc=Nc, steps=steps, iters=10, tops_burn=2, seed=5, scan=scan, code=cid, return_stats=True)
                ms = float(r["stats"]["kernel_ms"])
                row[scan] = dict(steps=steps, kernel_ms=ms, ladder_steps_per_s=N * steps / ms * 1e3)
            out["rows"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
