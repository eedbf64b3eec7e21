import sys, json
sys.path.insert(0, "mcmc-qec-toric-rl_amd")
import numpy as np
from qecmc import harness
for L in (5, 9):
    for crit, steps in (("error_based", 2000000), (None, 20000), (None, 200000)):
        out = harness.threshold_curve({"code": "toric", "size": L, "noise": "depolarizing"}, [0.05], 4096, seed=100 * L, steps=steps,
                                      conv_criteria=crit, device_generation=True)
        m = out["metrics"][0] or {}
        print(json.dumps(dict(L=L, criterion=crit, steps=steps, success=float(out["success_rate"][0]), err=float(out["err"][0]),
                              converged=float(out["converged_frac"][0]), mean_steps=m.get("mean_steps"))), flush=True)
