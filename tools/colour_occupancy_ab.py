"""Throughput of the colour-parallel layout (scan = "colour") for one build of the library: ladder steps per second at 256 ... 16 384 ladders.
    python tools/colour_occupancy_ab.py [--library path/to/libqecmc.so]      (one JSON line; tools/gpu_run.sh runs it once per build)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
from qecmc import _lib as L_
if "--library" in sys.argv:
    L_.use_library(sys.argv[sys.argv.index("--library") + 1])
import qecmc as q

rng = np.random.default_rng(3)
out = {"library": L_.library_path(), "rows": []}
for name, code, L, p, Nc, kw in (("toric L=9", q.TORIC, 9, 0.15, 8, {}), ("xzzx L=9 alpha", q.XZZX, 9, 0.175, 8, dict(alpha=4.04)), ("toric L=5", q.TORIC, 5, 0.1, 5, {})):
    shape = (2, L, L) if code == q.TORIC else (L, L)
    for N in (256, 1024, 4096, 16384):
        init = np.zeros((N,) + shape, dtype=np.uint8)
        err = rng.random(init.shape) < 0.12
        init[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
        for rep in range(2):
            r = q.pteq_batch(init, p, Nc=Nc, steps=1000, iters=10, tops_burn=2, seed=5, scan="colour", code=code, return_stats=True, **kw)
        ms = float(r["stats"]["kernel_ms"])
        out["rows"].append(dict(shape=name, ladders=N, kernel_ms=ms, ladder_steps_per_s=N * 1000 / ms * 1e3))
print(json.dumps(out))
