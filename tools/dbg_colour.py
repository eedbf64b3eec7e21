import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
import qecmc as q
from oracle import oracle as orc
rng = np.random.default_rng(1)
L = 5
init = (rng.integers(1, 4, size=(1, 2, L, L)) * (rng.random((1, 2, L, L)) < 0.1)).astype(np.uint8)
for Nc, pl, iters in ((1, 0.0, 1), (2, 0.0, 1), (3, 0.0, 1), (2, 1.0, 1), (3, 0.5, 3)):
    for steps in (1, 2, 3, 5, 20):
        got = q.pteq_batch(init, 0.1, Nc=Nc, steps=steps, iters=iters, tops_burn=0, seed=5, scan="colour", p_logical=pl, return_states=True)
        ld = orc.Ladder(orc.TORIC, init[0], 0.1, Nc, pl, scan=2)
        r = orc.Rng.philox(5, 0)
        for t in range(steps):
            ld.step(iters, r)
        same = [bool(np.array_equal(got["states"][0, c], ld.states[c])) for c in range(Nc)]
        print("Nc", Nc, "pl", pl, "iters", iters, "steps", steps, "rungs equal", same, "tops0", int(got["tops0"][0]), ld.tops0,
              "n gpu", [int(np.count_nonzero(got["states"][0, c])) for c in range(Nc)], "n orc", [int(np.count_nonzero(ld.states[c])) for c in range(Nc)])
