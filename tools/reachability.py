"""Which of the built kernels a caller can reach with production parameters (params.flags = 0), by running them: a sweep of tiny launches over
codes, lattice sizes, ladder lengths, rules, scan modes, fixed-length / criterion runs, replicas, the step and estimator entry points --
under `rocprofv3 --kernel-trace --stats`, whose kernel list is then set against the kernels the build instantiated (csrc/build/*.res).

    GPU box:  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/reach -- python3 tools/reachability.py sweep
    here:     python tools/reachability.py table gpurun_out/reach/*/*_kernel_stats.csv > profiles/r04_reachability.json
"""
import csv
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def sweep():
    import qecmc as q
    rng = np.random.default_rng(1)
    n_ok = n_ref = 0

    def run(fn, *a, **kw):
        nonlocal n_ok, n_ref
        try:
            fn(*a, **kw)
            n_ok += 1
        except q.QecmcError:
            n_ref += 1

    sizes = {"toric": [3, 4, 5, 7, 9, 10, 11, 12, 13, 15, 16, 17, 21], "planar": [3, 5, 7, 9, 11, 13, 16], "xzzx": [3, 5, 7, 9, 11, 13, 17, 21, 25, 31],
             "rotated": [3, 5, 7, 9, 11, 13, 17, 21, 25, 31]}
    codes = {"toric": q.TORIC, "planar": q.PLANAR, "xzzx": q.XZZX, "rotated": q.ROTATED}
    for name, cid in codes.items():
        for L in sizes[name]:
            shape = (2, L, L) if name in ("toric", "planar") else (L, L)
            for N in (3, 70):
                init = (rng.integers(1, 4, size=(N,) + shape) * (rng.random((N,) + shape) < 0.1)).astype(np.uint8)
                if name == "planar":
                    init[:, 1, -1, :] = 0; init[:, 1, :, -1] = 0
                for Nc in (1, 2, 3, 4, 5, 7, 8, 9, 12, 16):
                    rules = [dict()] + ([dict(eta=10.0), dict(alpha=2.0)] if name in ("xzzx", "rotated") else [])
                    for rule in rules:
                        for scan in ("random", "sweep", "colour", "wave"):
                            for iters in (10, 7):
                                base = dict(Nc=Nc, code=cid, scan=scan, iters=iters, tops_burn=0, seed=1, steps=3, **rule)
                                if N == 70 or scan != "wave":
                                    run(q.pteq_batch, init, 0.1, **base)
                                    run(q.pteq_batch, init, 0.1, return_states=True, **base)
                                    run(q.pteq_batch, init, 0.1, conv_criteria="error_based", **dict(base, steps=40))
                                    run(q.pteq_batch, init, 0.1, conv_criteria="error_based", return_states=True, **dict(base, steps=40))
                                    if iters == 10:
                                        run(q.pteq_batch, init, 0.1, replicas=2, **base)
                                        run(q.pteq_batch, init, 0.1, p_logical=0.0, **base)
                                        if scan == "random":
                                            run(q.pteq_batch, init, 0.1, return_swap_stats=True, **base)
            # the drop-in step calls and the unique-chain estimators (the batch form: one representative per class) on one small instance per size
            from qecmc import harness
            raw = (rng.integers(1, 4, size=(2,) + shape) * (rng.random((2,) + shape) < 0.1)).astype(np.uint8)
            if name == "planar":
                raw[:, 1, -1, :] = 0; raw[:, 1, :, -1] = 0
            code = {"toric": q.Toric_code, "planar": q.Planar_code, "xzzx": q.xzzx_code, "rotated": q.RotSurCode}[name](L)
            code.qubit_matrix = raw[0].copy()
            reps = harness.class_representatives(name, raw)
            for Nc in (1, 3, 8, 12):
                run(lambda: q.Ladder(0.1, code, Nc, 0.5, seed=1).step(10))
                for kw in (dict(), dict(per_rung=True), dict(with_m=True), dict(conv_mult=2.0, return_steps=True)):
                    run(q.ptdc_batch, reps, 0.2, Nc=Nc, steps=30, droplets=2, code=cid, **kw)
            run(lambda: q.Chain(0.1, code, seed=1).update_chain(10))
            run(q.ptdc_batch, reps, 0.2, Nc=1, steps=30, droplets=2, iters=5, code=cid)
            if name in ("xzzx", "rotated"):
                run(q.ptdc_batch, reps, 0.2, Nc=1, steps=30, droplets=2, iters=5, code=cid, alpha=2.0, with_xyz=True)
                run(lambda: q.Ladder_biased(0.1, code, 10.0, 3, 0.5, seed=1).step(10))
                run(lambda: q.Ladder_alpha(0.1, code, 2.0, 3, 0.5, seed=1).step(10))
            if name == "planar":
                run(q.ptdc_batch, reps, (0.05, 0.03, 0.04), Nc=1, steps=30, droplets=2, iters=5, code=cid, with_xyz=True)
    print(json.dumps({"launch_sets_run": n_ok, "refused": n_ref}))


def demangled_label(name):
    """'void qecmc::ladder_kernel<512, 8, 0, 1282u>(qecmc::LadderArgs)' -> the label tools/kernel_resources.py gives the mangled name"""
    from kernel_resources import CODES, FLAGS
    m = re.search(r"ladder_wu_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+), (true|false)>", name)
    if m:
        maxt, minw, code, wv, it = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(7))
        return "wave<%d,%d,%s: %d words%s%s%s%s>" % (maxt, minw, CODES[code], wv, ", conv" if m.group(5) == "true" else "", ", queue" if m.group(6) == "true" else "",
                                                   ", alpha" if m.group(8) == "true" else "", ", iters %d" % it if it else "")
    m = re.search(r"ladder_kernel<(\d+), (\d+), (\d+), (\d+)u>", name)
    if m:
        maxt, minw, code, fl = (int(x) for x in m.groups())
        return "ladder<%d,%d,%s: %s>" % (maxt, minw, CODES[code], "|".join(n for i, n in enumerate(FLAGS) if fl >> i & 1) or "plain")
    m = re.search(r"qecmc::([A-Za-z_0-9]+)", name)
    return (m.group(1) + re.sub(r".*?(<.*>)?\(.*", r"\1", name)) if m else name


def table(stats_csv):
    from kernel_resources import all_rows
    built = {}
    for r in all_rows():
        built.setdefault(r["label"], []).append(r)
    seen = {}
    for row in csv.DictReader(open(stats_csv)):
        if "qecmc::" in row["Name"]:
            lab = demangled_label(row["Name"])
            seen[lab] = seen.get(lab, 0) + int(row["Calls"])
    fam = lambda lab: lab.split("<")[0]
    ladder_built = [l for l in built if fam(l) in ("ladder", "wave")]
    out = {"what": __doc__.split("\n")[0], "kernels_built": sum(len(v) for v in built.values()), "ladder_and_wave_kernels_built": len(ladder_built),
           "ladder_and_wave_kernels_launched_by_the_sweep": sorted(l for l in ladder_built if l in seen),
           "ladder_and_wave_kernels_not_launched_by_the_sweep": sorted(l for l in ladder_built if l not in seen),
           "other_kernels_launched": sorted(l for l in seen if fam(l) not in ("ladder", "wave"))}
    out["counts"] = {"launched": len(out["ladder_and_wave_kernels_launched_by_the_sweep"]), "not_launched": len(out["ladder_and_wave_kernels_not_launched_by_the_sweep"])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1:2] == ["sweep"]:
        sweep()
    elif sys.argv[1:2] == ["table"]:
        table(sys.argv[2])
    else:
        print(__doc__)
