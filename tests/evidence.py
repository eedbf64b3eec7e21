#!/usr/bin/env python3
"""Full-size evidence runs of the BASELINE configurations on one MI355X (results -> gpurun_out/<tag>_*.json, copied to
profiles/).  Prints a progress line per stage so that a long run is never silent.

    python tests/evidence.py headline [--steps 100000] [--oracle 1024]     config 2 at the SURVEY 8d headline length
    python tests/evidence.py cfg3 [--steps 10000]                          config 3's per-GPU shard, full size, with property checks
    python tests/evidence.py cfg5 [--sweeps 1e5] [--syndromes 32768]       config 5: long-chain convergence study
    python tests/evidence.py cfg3long [--sweeps 1e5] [--syndromes 16384]   config 3's shape (toric L=15 p=0.18) run long: when do the ladders mix
    python tests/evidence.py threshold [--steps 300000] [--syndromes 16384] the harness end to end: success rate against p_error, L = 5, 7, 9
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
OUT = os.path.join(ROOT, "gpurun_out")


def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def binom(k, n):
    r = k / n
    return r, float(np.sqrt(max(r * (1 - r), 0.0) / n))


def headline(a):
    """config 2 at S = 1e5 ladder steps -- tops0 per syndrome, the fraction with tops0 >= 10, the aggregate
    argmax-success rate against the true class (observable 4), and the same through the oracle on a sub-sample."""
    import bench
    import qecmc
    from qecmc import toric_model as tm
    from oracle import oracle as orc
    N, L, p, Nc, S = 65536, 9, 0.15, 8, a.steps
    init, raw = bench.synth_batch(N, L, p, 20200915, return_raw=True)
    eq_true = np.asarray(tm.eq_class(raw))
    say("headline: batch ready, launching", N, "syndromes x", S, "ladder steps")
    res = qecmc.pteq_batch(init, p, Nc=Nc, steps=S, iters=10, tops_burn=2, seed=20200915, return_stats=True)
    k_ms = res["stats"]["kernel_ms"]
    proposals = N * Nc * 10 * S
    say("headline: kernel %.1f ms, %.3e proposals/s" % (k_ms, proposals / k_ms * 1e3))
    has = res["samples"] > 0
    ok = (np.argmax(res["counts"], axis=1) == eq_true)
    rec = {"workload": "configs[1]: toric L=9 p=0.15, 65 536 syndromes, Nc=8, iters=10, %d ladder steps (%.3g proposals, %.1f chain-sweeps per chain), tops_burn=2, seed 20200915" % (S, proposals, S * 10 / 162),
           "kernel_ms": k_ms, "proposals_per_s": proposals / k_ms * 1e3, "chain_sweeps_per_s": proposals / 162 / k_ms * 1e3,
           "roofline_frac_algorithmic": (proposals * 8 + N * (162 + 64)) / (k_ms * 1e-3) / 8e12,
           "tops0": {"mean": float(res["tops0"].mean()), "median": float(np.median(res["tops0"])), "max": int(res["tops0"].max()),
                     "hist_0_to_20plus": np.bincount(np.minimum(res["tops0"], 20).astype(np.int64), minlength=21).tolist(),
                     "frac_ge_2_past_burn_in": float(np.mean(res["tops0"] >= 2)), "frac_ge_10": float(np.mean(res["tops0"] >= 10))},
           "frac_syndromes_with_samples": float(has.mean())}
    r, e = binom(int(ok[has].sum()), int(has.sum()))
    rec["argmax_success_vs_true_class"] = {"rate_among_past_burn_in": r, "binomial_err": e, "n": int(has.sum()),
                                           "rate_all_syndromes_argmax_of_zeros_counts_as_class_0": float(ok.mean())}
    r10 = res["tops0"] >= 10
    if r10.any():
        r, e = binom(int(ok[r10].sum()), int(r10.sum()))
        rec["argmax_success_vs_true_class"]["rate_among_tops0_ge_10"] = r
        rec["argmax_success_vs_true_class"]["binomial_err_tops0_ge_10"] = e
    # a smaller run with the equilibrium observables (their LDS counters cost a workgroup of occupancy, so not in the timed launch)
    n_st = 4096
    st = qecmc.pteq_batch(init[:n_st], p, Nc=Nc, steps=min(S, 20000), iters=10, tops_burn=2, seed=20200915, return_swap_stats=True)
    rec["equilibrium_observables_first_4096_syndromes_%d_steps" % min(S, 20000)] = {
        "swap_acceptance_per_pair": (st["swap_accepts"].sum(axis=0) / (n_st * min(S, 20000))).tolist(),
        "mean_errors_per_rung": (st["nerr_sums"].sum(axis=0) / (n_st * min(S, 20000))).tolist(),
        "reference_F5_same_shape_3_syndromes": "swap acceptance [0.58 0.42 0.16 0.017 0.064 0.095 0.088], <n> [20.9 22.4 26.0 38.5 71.7 92.1 107.4 121.5] (tests/golden/f5_stats.npz)"}
    n_o = a.oracle
    if n_o:
        say("headline: oracle on", n_o, "syndromes,", os.cpu_count(), "threads")
        t0 = time.time()
        ref = orc.toric_pteq_batch(init[:n_o], p, Nc, S, iters=10, tops_burn=2, seed=20200915, n_threads=os.cpu_count() or 1)
        dt = time.time() - t0
        same = bool(np.array_equal(ref["counts"], res["counts"][:n_o]) and np.array_equal(ref["tops0"].astype(np.uint32), res["tops0"][:n_o]))
        ho = ref["samples"] > 0
        oko = np.argmax(ref["counts"], axis=1) == eq_true[:n_o]
        ro, eo = binom(int(oko[ho].sum()), max(int(ho.sum()), 1))
        rg, eg = binom(int(ok[:n_o][has[:n_o]].sum()), max(int(has[:n_o].sum()), 1))
        rec["oracle_subsample"] = {"syndromes": n_o, "seconds": dt, "cores": os.cpu_count(), "proposals_per_s": n_o * Nc * 10 * S / dt,
                                   "class_counts_and_tops0_bit_identical_to_gpu": same,
                                   "oracle_success_rate": ro, "oracle_err": eo, "gpu_success_rate_same_syndromes": rg, "gpu_err": eg}
        say("headline: oracle done in %.1f s, bit-identical: %s" % (dt, same))
    return rec


def cfg3(a):
    import bench
    import qecmc
    from qecmc import toric_model as tm
    from oracle import oracle as orc
    N, L, p, Nc, S = 131072, 15, 0.18, 8, a.steps
    init, raw = bench.synth_batch(N, L, p, 31, return_raw=True)
    eq_true = np.asarray(tm.eq_class(raw))
    say("cfg3: launching", N, "syndromes (one GPU's shard of the 1M batch) x", S, "ladder steps")
    res = qecmc.pteq_batch(init, p, Nc=Nc, steps=S, iters=10, tops_burn=2, seed=31, return_stats=True, return_states=True)
    k_ms = res["stats"]["kernel_ms"]
    proposals = N * Nc * 10 * S
    say("cfg3: kernel %.1f ms, %.3e proposals/s; checking" % (k_ms, proposals / k_ms * 1e3))
    syn0 = tm.syndrome(init)
    conserved = all(bool(np.array_equal(tm.syndrome(np.ascontiguousarray(res["states"][:, c])), syn0)) for c in range(Nc))
    sums = bool(np.array_equal(res["counts"].sum(axis=1), res["samples"]))
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(N, size=32, replace=False))
    same = True
    for s in pick:
        ref = orc.toric_pteq_batch(init[s:s + 1], p, Nc, S, iters=10, tops_burn=2, seed=31, first_syndrome=int(s), return_states=True)
        same &= bool(np.array_equal(ref["counts"][0], res["counts"][s]) and np.array_equal(ref["states"][0], res["states"][s]))
    has = res["samples"] > 0
    ok = np.argmax(res["counts"], axis=1) == eq_true
    rec = {"workload": "configs[2], one GPU's shard: toric L=15 p=0.18, 131 072 syndromes, Nc=8, iters=10, %d ladder steps (%.3g proposals)" % (S, proposals),
           "kernel_ms": k_ms, "proposals_per_s": proposals / k_ms * 1e3, "chain_sweeps_per_s": proposals / 450 / k_ms * 1e3,
           "roofline_frac_algorithmic": (proposals * 8 + N * (450 + 64)) / (k_ms * 1e-3) / 8e12,
           "properties": {"syndrome_of_every_rung_of_every_ladder_conserved": conserved, "class_counts_sum_to_samples": sums,
                          "random_32_syndromes_bit_identical_to_oracle_counts_and_final_states": bool(same)},
           "tops0": {"mean": float(res["tops0"].mean()), "frac_ge_2": float(np.mean(res["tops0"] >= 2)), "frac_ge_10": float(np.mean(res["tops0"] >= 10))},
           "argmax_success_among_past_burn_in": binom(int(ok[has].sum()), max(int(has.sum()), 1)) + (int(has.sum()),)}
    return rec


def cfg5(a, name="rotated", L=21, p=0.17, Nc=8, label="configs[4] on one GPU: rotated L=21 p=0.17", tag="cfg5"):
    import qecmc
    from qecmc import harness
    N = a.syndromes
    qcode = {"rotated": qecmc.ROTATED, "toric": qecmc.TORIC}[name]
    G = L * L - 1 if name == "rotated" else 2 * L * L
    total_steps = int(round(a.sweeps * G / 10))                      # ladder steps of iters=10 for `sweeps` sweeps per chain
    rng = np.random.default_rng(5)
    raw = harness.draw_errors(name, L, N, p, rng)
    true = np.asarray(harness._class_of(qcode, raw))
    init = harness.hide_class(name, raw, rng)
    sw = [s for s in (10, 30, 100, 300, 1e3, 3e3, 1e4, 3e4, 1e5, 3e5, 1e6) if s < a.sweeps] + [a.sweeps]
    cps = [int(round(s * G / 10)) for s in sw]
    run = harness.LadderRun(init, p, Nc=Nc, iters=10, tops_burn=2, seed=5, code=qcode)
    rows, t0 = [], time.time()
    chunk = 1 << 19
    for c in cps:
        while run.steps < c:
            run.advance(min(chunk, c - run.steps))
            run._torch.cuda.synchronize()
            say("%s: %d / %d ladder steps (%.1f s)" % (tag, run.steps, total_steps, time.time() - t0))
        s = run.snapshot()
        has = s["samples"] > 0
        ok = np.argmax(s["counts"], axis=1) == true
        frac = s["counts"] / np.maximum(s["samples"], 1)[:, None].astype(np.float64)
        rows.append(dict(ladder_steps=int(run.steps), sweeps_per_chain=run.steps * 10 / G, wall_s=time.time() - t0,
                         frac_past_burn_in=float(has.mean()), mean_tops0=float(s["tops0"].mean()), frac_tops0_ge_10=float(np.mean(s["tops0"] >= 10)),
                         argmax_success_past_burn_in=float(ok[has].mean()) if has.any() else None,
                         mean_prob_of_true_class=float(frac[np.arange(N), true][has].mean()) if has.any() else None, _frac=frac))
    dt = time.time() - t0
    last = rows[-1]["_frac"]
    for r in rows:
        r["mean_tv_distance_to_last_checkpoint"] = float(0.5 * np.abs(r.pop("_frac") - last).sum(axis=1).mean())
    proposals = float(run.steps) * N * Nc * 10
    run.close()
    return {"workload": "%s Nc=%d iters=10, %d syndromes (class hidden by a random logical), %.3g sweeps per chain = %d ladder steps, exact chunked continuation (LadderRun, %d-step launches)" % (label, Nc, N, a.sweeps, total_steps, chunk),
            "checkpoints": rows, "wall_s": dt, "proposals": proposals, "proposals_per_s_wall": proposals / dt,
            "chain_sweeps_per_s_wall": proposals / G / dt, "roofline_frac_algorithmic_wall": proposals * 8 / dt / 8e12}


def cfg3long(a):
    """Config 3's shape run long: at 10^4 ladder steps (the bench length) almost no toric L=15 ladder at p=0.18 has passed the tops0 burn-in."""
    return cfg5(a, name="toric", L=15, p=0.18, Nc=8, label="configs[2]'s shape run long on one GPU: toric L=15 p=0.18", tag="cfg3long")


def threshold(a):
    """The reference's data-generation recipe end to end (generate_data.py:57-60,110-141,276-296), batched: for toric L = 5, 7, 9 and
    p_error over [0.05, 0.20], `--syndromes` errors drawn on the GPU, their class hidden by a random logical operator, decoded by PTEQ with the
    reference's defaults (Nc = L, error_based criterion SEQ = 2 / TOPS = 10 / eps = 0.1, horizon `--steps` ladder steps); success =
    argmax(distribution) == true class.  The curves of different L cross near the code's threshold for depolarizing noise."""
    from qecmc import harness
    ps = [0.05, 0.08, 0.11, 0.13, 0.15, 0.16, 0.17, 0.18, 0.19, 0.20]
    rec = {"workload": "toric L = 5, 7, 9, depolarizing noise, %d syndromes per (L, p) generated on the device, PTEQ defaults (Nc = L, error_based "
                       "criterion, horizon %d ladder steps)" % (a.syndromes, a.steps), "p": ps, "curves": {}}
    for L in (5, 7, 9):
        t0 = time.time()
        out = harness.threshold_curve({"code": "toric", "size": L, "noise": "depolarizing"}, ps, a.syndromes, seed=100 * L, steps=a.steps,
                                      conv_criteria="error_based", device_generation=True)
        rec["curves"]["L%d" % L] = dict(success_rate=[float(x) for x in out["success_rate"]], err=[float(x) for x in out["err"]],
                                        converged_frac=[float(x) for x in out["converged_frac"]],
                                        # among the syndromes whose ladder got past the burn-in at all (the reference returns the all-zero
                                        # vector -> class 0 for the others, decoders.py:63,89: the raw rate at low p measures that trap too)
                                        success_rate_sampled=[float(x) for x in out["success_rate_sampled"]],
                                        err_sampled=[float(x) for x in out["err_sampled"]], frac_sampled=[float(x) for x in out["frac_sampled"]],
                                        mean_steps=[float(m["mean_steps"]) if m and "mean_steps" in m else None for m in out["metrics"]],
                                        wall_s=time.time() - t0)
        say("threshold: L=%d done in %.1f s: " % (L, time.time() - t0) + " ".join("%.3f" % x for x in out["success_rate"]))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["headline", "cfg3", "cfg5", "cfg3long", "threshold"])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--oracle", type=int, default=1024)
    ap.add_argument("--sweeps", type=float, default=1e5)
    ap.add_argument("--syndromes", type=int, default=32768)
    ap.add_argument("--tag", default="r03")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 100000 if a.what == "headline" else 10000
    if a.what == "threshold" and a.syndromes == 32768:
        a.syndromes = 16384
    rec = {"headline": headline, "cfg3": cfg3, "cfg5": cfg5, "cfg3long": cfg3long, "threshold": threshold}[a.what](a)
    os.makedirs(OUT, exist_ok=True)
    name = {"headline": "%s_headline_S%g.json" % (a.tag, a.steps), "cfg3": "%s_cfg3_full_S%g.json" % (a.tag, a.steps),
            "cfg5": "%s_cfg5_convergence_%gsweeps.json" % (a.tag, a.sweeps), "threshold": "%s_threshold_curves.json" % a.tag,
            "cfg3long": "%s_cfg3_convergence_%gsweeps.json" % (a.tag, a.sweeps)}[a.what]
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(rec, f, indent=1)
    say("wrote", name)
    print(json.dumps(rec)[:3000])


if __name__ == "__main__":
    main()
