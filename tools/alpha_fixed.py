"""Fixed-length launches of the alpha ladder (PTEQ_alpha's chain without the criterion): scan = random against scan = wave, kernel time.
    python tools/alpha_fixed.py L Nc N steps"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
import qecmc
from qecmc import harness
L, Nc, N, steps = (int(v) for v in sys.argv[1:5])
rng = np.random.default_rng(7)
raw = harness.draw_errors(qecmc.XZZX, L, N, 0.15, rng, eta=100.0)
init = harness.hide_class(qecmc.XZZX, raw, rng)
p_dec, a = harness.biased_as_alpha(0.15, 100.0)
out = {}
# ("wave+criterion": the criterion kernel on its queue with a criterion that never holds (eps = 0): every ladder runs to the horizon, one per lane --
#  what the bookkeeping of a criterion run costs per step)
for scan in ("random", "wave", "wave+criterion", "random", "wave", "wave+criterion"):
    kw = dict(conv_criteria="error_based", eps=0.0) if scan.endswith("criterion") else {}
    r = qecmc.pteq_batch(init, p_dec, Nc=Nc, steps=steps, iters=10, tops_burn=2, seed=3, return_stats=True, scan=scan.split("+")[0], code=qecmc.XZZX, alpha=float(a), **kw)
    ms = r["stats"]["kernel_ms"]
    out[scan] = dict(kernel_ms=ms, proposals_per_s=N * Nc * steps * 10 / (ms * 1e-3), roofline_frac=N * Nc * steps * 10 / (ms * 1e-3) * 8 / 8e12)
    print(scan, json.dumps(out[scan]), flush=True)
