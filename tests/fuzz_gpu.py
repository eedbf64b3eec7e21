"""Randomised parity sweep (run by hand on the GPU box: `python tests/fuzz_gpu.py [cases] [seed]`): draws ladder
configurations at random -- code, lattice size, number of rungs, proposals per step, noise model, scan mode, fixed length or the
convergence criterion, batch sizes that leave ragged workgroups -- runs `pteq_batch` on the GPU and the CPU oracle on the same
Philox streams, and demands identical class counts, sample counts, tops0, stopping steps and (fixed-length runs) final states
of every rung.  The oracle is the checker here, as in tests/ proper; nothing under oracle/ is on the product path.
Writes gpurun_out/fuzz_<seed>.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))
sys.path.insert(0, ROOT)
import qecmc as q                        # noqa: E402
from oracle import oracle as orc         # noqa: E402


def draw_case(rng):
    code = rng.choice(["toric", "toric", "xzzx", "rotated", "planar"])
    if code == "toric":
        L = int(rng.choice([3, 4, 5, 6, 7, 9, 10, 12, 13, 15, 16, 17]))
    elif code == "planar":
        L = int(rng.choice([3, 4, 5, 7, 9]))
    else:
        L = int(rng.choice([3, 5, 7, 9, 11, 13, 17, 21]))
    Nc = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16]))
    nq = 2 * L * L if code in ("toric", "planar") else L * L
    while Nc > 1 and (Nc * ((nq + 15) // 16) * 256 + Nc * 1100 + 16 * nq * 2 + 12000 > 150 * 1024):
        Nc -= 1                                                            # keep the workgroup inside the LDS
    noise = "depolarizing"
    if code in ("xzzx", "rotated") and nq <= 400 and rng.random() < 0.35:
        noise = rng.choice(["biased", "alpha"])
    scan = "sweep" if (noise == "depolarizing" and rng.random() < 0.15) else "random"
    conv = noise != "alpha" and rng.random() < 0.3
    iters = int(rng.choice([1, 2, 3, 5, 7, 8, 10, 10, 10, 12, 13, 25]))
    work = Nc * iters * nq                                                 # ~ oracle cost per ladder step (the stencil copies nq bytes)
    steps = int(max(3, min(400 if not conv else 2500, 6e6 // work)))
    N = int(rng.choice([1, 7, 33, 64, 65, 100, 130]))
    N = max(1, min(N, int(2e8 // (work * steps)) + 1))
    p = float(rng.choice([0.05, 0.1, 0.15, 0.18, 0.3]))
    # runs that stop by the criterion: a persistent grid of one or two workgroups makes finished lanes take new ladders from the
    # queue; fixed-length random-scan runs: sometimes cut into chunks continued from device-resident state (harness.LadderRun)
    grid = str(rng.choice(["", "1", "2"])) if conv else ""
    chunks = noise != "alpha" and scan == "random" and not conv and rng.random() < 0.25
    return dict(code=code, L=L, Nc=Nc, noise=noise, scan=scan, conv=conv, iters=iters, steps=steps, N=N, p=p, grid=grid, chunks=bool(chunks),
                tops_burn=int(rng.choice([0, 1, 2])), seed=int(rng.integers(1, 1 << 30)), first=int(rng.integers(0, 1000)),
                eta=float(rng.choice([3.0, 10.0, 100.0])), alpha=float(rng.choice([1.3, 2.0, 3.1])))


def run_case(c, rng):
    code = c["code"]
    shape = (c["N"], 2, c["L"], c["L"]) if code in ("toric", "planar") else (c["N"], c["L"], c["L"])
    init = np.zeros(shape, np.uint8)
    err = rng.random(shape) < c["p"]
    init[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    qcode = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED, "planar": q.PLANAR}[code]
    ocode = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED, "planar": orc.PLANAR}[code]
    kw = dict(steps=c["steps"], iters=c["iters"], tops_burn=c["tops_burn"], seed=c["seed"], first_syndrome=c["first"])
    if c["conv"]:
        kw.update(conv_criteria="error_based", SEQ=1, TOPS=3, eps=0.6)
    gk = dict(kw, code=qcode, Nc=c["Nc"], scan=c["scan"], return_states=not c["conv"])
    ok = dict(kw, return_states=not c["conv"], scan=1 if c["scan"] == "sweep" else 0)
    p = c["p"]
    if c["noise"] == "biased":
        gk["eta"] = c["eta"]; ok.update(noise=orc.BIASED, eta=c["eta"])
    elif c["noise"] == "alpha":
        gk["alpha"] = c["alpha"]; ok.update(noise=orc.ALPHA, alpha=c["alpha"], det_pow=1)
    steps = ok.pop("steps")
    if c["grid"]:
        os.environ["QECMC_QUEUE_GRID"] = c["grid"]
    else:
        os.environ.pop("QECMC_QUEUE_GRID", None)
    if c["chunks"]:
        from qecmc import harness
        run = harness.LadderRun(init, p, Nc=c["Nc"], iters=c["iters"], tops_burn=c["tops_burn"], seed=c["seed"], first_syndrome=c["first"],
                                code=qcode, eta=c["eta"] if c["noise"] == "biased" else None)
        left = steps
        while left > 0:
            n = int(min(left, rng.integers(1, max(2, steps // 2 + 1))))
            run.advance(n); left -= n
        got = run.snapshot(states=True)
        got["steps_done"] = np.full(c["N"], steps); got["converged"] = np.zeros(c["N"], bool)
        run.close()
    else:
        got = q.pteq_batch(init, p, **gk)
    ref = orc.pteq_batch(ocode, init, p, c["Nc"], steps, **ok)
    bad = []
    for key in ("counts", "samples", "tops0", "steps_done", "converged"):
        if not np.array_equal(np.asarray(got[key]).astype(np.uint64), np.asarray(ref[key]).astype(np.uint64)):
            bad.append(key)
    if not c["conv"] and not np.array_equal(got["states"], ref["states"]):
        bad.append("states")
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    failures, done, kinds = [], 0, {}
    for i in range(cases):
        c = draw_case(rng)
        try:
            bad = run_case(c, rng)
        except q.QecmcError as e:                                          # a shape the library refuses (LDS, table size): said so, fine
            bad = None
            kinds["refused"] = kinds.get("refused", 0) + 1
            if "unsupported" not in str(e).lower() and "lds" not in str(e).lower() and "exceed" not in str(e).lower():
                failures.append(dict(case=c, error=str(e)))
        if bad:
            failures.append(dict(case=c, differs=bad))
            print("MISMATCH", c, bad, flush=True)
        if bad is not None:
            done += 1
            k = "%s/%s/%s%s%s%s" % (c["code"], c["noise"], c["scan"], "/conv" if c["conv"] else "", "/queue-grid-" + c["grid"] if c["grid"] else "",
                                    "/chunked" if c["chunks"] else "")
            kinds[k] = kinds.get(k, 0) + 1
        if i % 20 == 19:
            print("%d cases, %d compared, %d failures, %.0f s" % (i + 1, done, len(failures), time.time() - t0), flush=True)
    out = dict(seed=seed, cases=cases, compared=done, failures=failures, kinds=kinds, seconds=time.time() - t0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fuzz_%d.json" % seed), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "failures"}))
    print("FAILURES: %d" % len(failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
