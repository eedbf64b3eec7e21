"""Single-syndrome latency: wall time for ONE syndrome -- the reference's real call pattern,
decoders.py:25, generate_data.py:136 -- to reach the reference's default stop (error_based: SEQ=2, TOPS=10, eps=0.1) and to reach
tops0 >= 10, in the lane-per-chain layout (replicas = 1 and 64) and, where built, the colour-parallel layout (scan = "colour").
GPU box:  python tools/latency.py > gpurun_out/r03_latency.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
import qecmc as q
import bench


def first_step_with_tops(init, p, Nc, want, **kw):
    """smallest run length (to a factor 1.25) whose tops0 reaches `want`, and the kernel / wall time of that run (Philox: a shorter run is
    the exact prefix of a longer one)"""
    steps = 2000
    while True:
        t0 = time.perf_counter()
        r = q.pteq_batch(init, p, Nc=Nc, steps=steps, iters=10, tops_burn=2, seed=11, return_stats=True, **kw)
        wall = time.perf_counter() - t0
        if int(r["tops0"].min()) >= want * max(int(kw.get("replicas", 1)), 1) or steps >= 4_000_000:
            return dict(steps=steps, tops0=int(r["tops0"][0]), kernel_ms=float(r["stats"]["kernel_ms"]), wall_ms=wall * 1e3)
        steps = int(steps * 1.25)


def default_stop(init, p, Nc, **kw):
    horizon = 1 << 16
    tot_k, t0 = 0.0, time.perf_counter()
    while True:
        r = q.pteq_batch(init, p, Nc=Nc, steps=horizon, iters=10, tops_burn=2, seed=11, conv_criteria="error_based", return_stats=True, **kw)
        tot_k += float(r["stats"]["kernel_ms"])
        if r["converged"][0] or horizon >= (1 << 22):
            break
        horizon *= 4
    return dict(converged=bool(r["converged"][0]), steps_done=int(r["steps_done"][0]), kernel_ms_last=float(r["stats"]["kernel_ms"]),
                kernel_ms_all_horizons=tot_k, wall_ms=(time.perf_counter() - t0) * 1e3, argmax=int(np.argmax(r["counts"][0])))


def biased_rows():
    """--biased: config 4's shape as the reference decodes it, one syndrome per call (generate_data.py:136-150): xzzx L = 9, errors at p = 0.15, eta = 100,
    decoded by PTEQ_biased (eta) and by PTEQ_alpha with (pz_tilde, alpha) derived from them -- lane-per-chain against the colour phases"""
    from qecmc import harness
    res = {"note": "xzzx L=9, errors at p=0.15 eta=100, Nc=9 (the reference's default Nc = L); one syndrome per call; kernel_ms = the kernel alone"}
    rng = np.random.default_rng(7)
    raw = harness.draw_errors(q.XZZX, 9, 4, 0.15, rng, eta=100.0)
    inits = harness.hide_class(q.XZZX, raw, rng)
    pzt, al = harness.biased_as_alpha(0.15, 100.0)
    q.pteq_batch(inits[:1], 0.15, Nc=5, steps=10, code=q.XZZX, eta=100.0)
    for label, pdec, kw in (("PTEQ_biased(eta=100)", 0.15, dict(code=q.XZZX, eta=100.0)), ("PTEQ_alpha(pz_tilde=%.4g, alpha=%.4g)" % (pzt, al), float(pzt), dict(code=q.XZZX, alpha=float(al)))):
        rows = []
        for s in range(4):
            init = inits[s:s + 1]
            row = {"syndrome": s}
            for name, extra in (("lane_per_chain", {}), ("colour_parallel", dict(scan="colour"))):
                if name == "lane_per_chain" and "--only-colour" in sys.argv:
                    continue
                row[name] = {"to_tops0_ge_10": first_step_with_tops(init, pdec, 9, 10, **kw, **extra), "default_stop": default_stop(init, pdec, 9, **kw, **extra)}
            rows.append(row)
            print(label, json.dumps(row), file=sys.stderr, flush=True)
        res[label] = rows
    return res


if "--biased" in sys.argv:
    print(json.dumps(biased_rows(), indent=1))
    sys.exit(0)
out = {"note": "one syndrome per call; lane-per-chain layout = the reference's random scan (scan=random); times include the "
               "host-pointer boundary (H2D, launch, D2H) in wall_ms and the kernel alone in kernel_ms"}
q.pteq_batch(bench.synth_batch(1, 5, 0.1, 1), 0.1, Nc=5, steps=10)          # load the library, warm the context
ONLY_COLOUR = "--only-colour" in sys.argv          # (the lane-per-chain rows take minutes: profiles/r03_latency.json keeps them)
for name, L, p, Nc in (("toric L=9 p=0.15 Nc=8", 9, 0.15, 8), ("toric L=15 p=0.18 Nc=8", 15, 0.18, 8)):
    rows = []
    for s in range(4 if L == 9 else 2):
        init, raw = bench.synth_batch(1, L, p, 100 + s, return_raw=True)
        row = {"syndrome_seed": 100 + s}
        for R in (() if ONLY_COLOUR else (1, 64)):
            kw = dict(replicas=R) if R > 1 else {}
            row["replicas_%d" % R] = {"to_tops0_ge_10": first_step_with_tops(init, p, Nc, 10, **kw), "default_stop": default_stop(init, p, Nc, **kw)}
        if "colour" in getattr(q, "SCANS", ()):
            # (iters = 10 PHASES per step here: 10 / n_phases sweeps of every rung between swap sweeps, against 10 / G in the
            # lane-per-chain layout)
            row["colour_parallel"] = {"to_tops0_ge_10": first_step_with_tops(init, p, Nc, 10, scan="colour"),
                                      "default_stop": default_stop(init, p, Nc, scan="colour")}
        rows.append(row)
        print(name, json.dumps(row), file=sys.stderr, flush=True)
    out[name] = rows
print(json.dumps(out, indent=1))
