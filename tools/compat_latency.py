"""Per-call cost of the drop-in (N = 1) entry points -- `Chain.update_chain(iters)`, `Ladder.step(iters)` and the reference's PTEQ loop written
over them (decoders.py:55-71: one `ladder.step(iters)` per Python iteration) -- i.e. what a caller who changes nothing but the import pays.
GPU box:  python tools/compat_latency.py > gpurun_out/r03_compat_latency.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
import qecmc as q


def per_call(fn, n):
    fn()
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    return dict(calls=n, median_us=float(np.median(t)), p10_us=float(np.percentile(t, 10)), p90_us=float(np.percentile(t, 90)))


def main():
    out = {"what": __doc__.split("\n")[0], "device": q.device_name() if hasattr(q, "device_name") else None, "rows": []}
    rng = np.random.default_rng(5)
    for name, mk, L, p in (("toric", lambda L: q.Toric_code(L), 5, 0.10), ("toric", lambda L: q.Toric_code(L), 9, 0.15),
                           ("xzzx", lambda L: q.xzzx_code(L), 9, 0.15), ("rotated", lambda L: q.RotSurCode(L), 21, 0.17)):
        code = mk(L)
        code.generate_random_error(p) if name == "toric" else code.generate_random_error(p / 3, p / 3, p / 3)
        ch = q.Chain(p, code, seed=3)
        row = dict(code=name, L=L, p=p)
        row["Chain.update_chain(10)"] = per_call(lambda: ch.update_chain(10), 300)
        Nc = 8
        lad = q.Ladder(p, code, Nc, 0.5, seed=3)
        row["Ladder.step(10)"] = per_call(lambda: lad.step(10), 300)
        row["Ladder.step(10, nsteps=1000)"] = per_call(lambda: lad.step(10, 1000), 20)
        out["rows"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
