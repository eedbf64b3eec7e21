"""Randomised parity sweep (run by hand on the GPU box: `python tests/fuzz_gpu.py [cases] [seed]`): draws ladder
configurations at random -- code, lattice size, number of rungs, proposals per step, noise model, scan mode, fixed length or the
convergence criterion, batch sizes that leave ragged workgroups -- runs `pteq_batch` on the GPU and the CPU oracle on the same
Philox streams, and demands identical class counts, sample counts, tops0, stopping steps and (fixed-length runs) final states
of every rung.  The oracle is the checker here, as in tests/ proper; nothing under oracle/ is on the product path.
Writes gpurun_out/fuzz_<seed>.json.  `python tests/fuzz_gpu.py other [cases] [seed]` does the same for the other entry points
(chain updates, Ladder.step in chunks, the unique-chain estimators' ptdc_batch).""" 
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
    u = rng.random()
    scan = "random" if noise != "depolarizing" else "sweep" if u < 0.15 else "colour" if (u < 0.3 and Nc >= 2) else "random"
    # (round 4) the colour phases under the biased / alpha rules (Ladder_alpha needs its top rung: Nc >= 2)
    if noise != "depolarizing" and (noise == "biased" or Nc >= 2) and rng.random() < 0.3:
        scan = "colour"
    # (round 4) scan = wave where it is built: the depolarizing rule up to 16 state words per rung, the alpha rule up to 8 words and 8 rungs
    W = (nq + 15) // 16
    if Nc >= 2 and ((noise == "depolarizing" and W <= 16) or (noise == "alpha" and W <= 8)) and rng.random() < (0.6 if noise == "alpha" else 0.3):
        scan = "wave"
    conv = rng.random() < 0.3                             # (round 3: the alpha rule's criterion runs take the work queue too)
    iters = int(rng.choice([1, 2, 3, 5, 7, 8, 10, 10, 10, 12, 13, 25]))
    work = Nc * iters * nq                                                 # ~ oracle cost per ladder step (the stencil copies nq bytes)
    steps = int(max(3, min(400 if not conv else 2500, 6e6 // work)))
    N = int(rng.choice([1, 7, 33, 64, 65, 100, 130]))
    N = max(1, min(N, int(2e8 // (work * steps)) + 1))
    p = float(rng.choice([0.05, 0.1, 0.15, 0.18, 0.3]))
    # runs that stop by the criterion: a persistent grid of one or two workgroups makes finished lanes take new ladders from the
    # queue; fixed-length random-scan runs: sometimes cut into chunks continued from device-resident state (harness.LadderRun)
    grid = str(rng.choice(["", "1", "2"])) if (conv and scan != "colour") else ""
    chunks = noise != "alpha" and scan == "random" and not conv and rng.random() < 0.25
    # replica ladders (R per syndrome, results summed on the device) in some of the one-launch fixed-length runs
    R = int(rng.choice([2, 3, 5])) if (not conv and not chunks and noise != "alpha" and rng.random() < 0.2) else 1
    if R > 1:
        N = max(1, N // R)
    return dict(code=code, L=L, Nc=Nc, noise=noise, scan=scan, conv=conv, iters=iters, steps=steps, N=N, p=p, grid=grid, chunks=bool(chunks), R=R,
                tops_burn=int(rng.choice([0, 1, 2])), seed=int(rng.integers(1, 1 << 30)), first=int(rng.integers(0, 1000)) * (64 if scan == "wave" else 1),
                eta=float(rng.choice([3.0, 10.0, 100.0])), alpha=float(rng.choice([1.3, 2.0, 3.1])))


def run_case(c, rng):
    code = c["code"]
    shape = (c["N"], 2, c["L"], c["L"]) if code in ("toric", "planar") else (c["N"], c["L"], c["L"])
    init = np.zeros(shape, np.uint8)
    err = rng.random(shape) < c["p"]
    init[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    if code == "planar":
        init[:, 1, -1, :] = 0; init[:, 1, :, -1] = 0                       # (the idle row and column of layer 1)
    qcode = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED, "planar": q.PLANAR}[code]
    ocode = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED, "planar": orc.PLANAR}[code]
    kw = dict(steps=c["steps"], iters=c["iters"], tops_burn=c["tops_burn"], seed=c["seed"], first_syndrome=c["first"])
    if c["conv"]:
        kw.update(conv_criteria="error_based", SEQ=1, TOPS=3, eps=0.6)
    gk = dict(kw, code=qcode, Nc=c["Nc"], scan=c["scan"], return_states=not c["conv"])
    ok = dict(kw, return_states=not c["conv"], scan={"random": 0, "sweep": 1, "colour": 2, "wave": 3}[c["scan"]])
    p = c["p"]
    if c["noise"] == "biased":
        gk["eta"] = c["eta"]; ok.update(noise=orc.BIASED, eta=c["eta"])
    elif c["noise"] == "alpha":
        gk["alpha"] = c["alpha"]; ok.update(noise=orc.ALPHA, alpha=c["alpha"], det_pow=1)
    steps = ok.pop("steps")
    if c["grid"]:
        gk["flags"] = q.dev_flags(queue_grid=int(c["grid"]))
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
        got = q.pteq_batch(init, p, **gk, **({"replicas": c["R"]} if c["R"] > 1 else {}))
    if c["scan"] == "wave" and c["conv"]:
        # the criterion runs of scan = wave take the deterministic per-workgroup queue: the oracle's restatement of it, on the grid the launch uses
        groups = (c["N"] + 63) // 64
        g_eff = max(1, min(int(c["grid"]) if c["grid"] else groups, groups))
        wk = {k: v for k, v in ok.items() if k not in ("return_states", "scan", "conv_criteria")}
        ref = orc.pteq_wave_queue(ocode, init, p, c["Nc"], steps, g_eff, **wk)
    else:
        ref = orc.pteq_batch(ocode, np.repeat(init, c["R"], axis=0), p, c["Nc"], steps, **ok)
    if c["R"] > 1:                                                          # ladder l = s R + r: Philox index first + l, summed per syndrome
        ncls = ref["counts"].shape[1]
        ref = dict(ref, counts=ref["counts"].reshape(c["N"], c["R"], ncls).sum(axis=1), samples=ref["samples"].reshape(c["N"], c["R"]).sum(axis=1),
                   tops0=ref["tops0"].reshape(c["N"], c["R"]).sum(axis=1), steps_done=ref["steps_done"].reshape(c["N"], c["R"]).max(axis=1),
                   converged=ref["converged"].reshape(c["N"], c["R"]).all(axis=1))
    bad = []
    # (scan = colour reports the first step with tops0 >= TOPS in steps_done / converged: tests/test_gpu_colour.py checks those)
    for key in ("counts", "samples", "tops0") + (() if (c["scan"] == "colour" and not c["conv"]) else ("steps_done", "converged")):
        if not np.array_equal(np.asarray(got[key]).astype(np.uint64), np.asarray(ref[key]).astype(np.uint64)):
            bad.append(key)
    if not c["conv"] and not np.array_equal(got["states"], ref["states"]):
        bad.append("states")
    return bad


def run_other(rng):
    """The other entry points: Chain.update_chain / Chain_biased (one chain, caller's stream and proposal index), Ladder.step in
    chunks, and the unique-chain estimators' fused sampling + set-insertion launches (ptdc_batch: PTDC / STDC / PTRC / STRC)."""
    kind = str(rng.choice(["chain", "chain", "ladder", "ptdc"]))
    name = str(rng.choice(["toric", "xzzx", "rotated", "planar"]))
    L = int(rng.choice([3, 4, 5, 7, 9] if name in ("toric", "planar") else [3, 5, 7, 9, 13]))
    cls = {"toric": q.Toric_code, "xzzx": q.xzzx_code, "rotated": q.RotSurCode, "planar": q.Planar_code}[name]
    ocode = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED, "planar": orc.PLANAR}[name]
    qcode = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED, "planar": q.PLANAR}[name]
    shape = (2, L, L) if name in ("toric", "planar") else (L, L)
    m = (rng.integers(1, 4, size=shape) * (rng.random(shape) < 0.15)).astype(np.uint8)
    if name == "planar":
        m[1, -1, :] = 0; m[1, :, -1] = 0                                    # (the idle row and column of layer 1)
    seed, stream = int(rng.integers(1, 1 << 40)), int(rng.integers(0, 50))
    p = float(rng.choice([0.05, 0.12, 0.2, 0.4, 0.75]))
    desc = dict(kind=kind, code=name, L=L, p=p, seed=seed, stream=stream)
    if kind == "chain":
        eta = float(rng.choice([3.0, 100.0])) if (name in ("xzzx", "rotated") and rng.random() < 0.4) else None
        p_logical = float(rng.choice([0.0, 0.0, 0.25, 0.5, 1.0]))
        iters, slot, k0 = int(rng.integers(1, 600)), int(rng.integers(0, 16)), int(rng.integers(0, 100000))
        desc.update(eta=eta, p_logical=p_logical, iters=iters, slot=slot, k0=k0)
        code = cls(L); code.qubit_matrix = m.copy()
        ch = q.Chain(p, code, seed=seed, stream=stream) if eta is None else q.Chain_biased(min(p, 0.4), eta, code, seed=seed, stream=stream)
        ch.p_logical, ch.slot, ch.proposals_done = p_logical, slot, k0
        ch.update_chain(iters)
        ref = orc.chain_update(ocode, m, p if eta is None else min(p, 0.4), p_logical, iters, orc.Rng.philox(seed, stream), slot=slot, k0=k0,
                               noise=0 if eta is None else 1, eta=eta or 0.0)
        return desc, ([] if np.array_equal(ch.code.qubit_matrix, ref) else ["state"])
    if kind == "ladder":
        Nc, iters = int(rng.choice([1, 2, 3, 5, 8, 16])), int(rng.choice([1, 3, 5, 10, 13]))
        pb = min(p, 0.3)
        desc.update(Nc=Nc, iters=iters, p=pb)
        code = cls(L); code.qubit_matrix = m.copy()
        ld = q.Ladder(pb, code, Nc, 0.5, seed=seed, stream=stream)
        ref = orc.Ladder(ocode, m, pb, Nc, 0.5)
        r = orc.Rng.philox(seed, stream)
        bad = []
        for chunk in (1, int(rng.integers(1, 5)), int(rng.integers(1, 30))):
            ld.step(iters, nsteps=chunk)
            for _ in range(chunk):
                ref.step(iters, r)
            got = np.stack([c.code.qubit_matrix for c in ld.chains])
            if not np.array_equal(got, ref.states) or [c.flag for c in ld.chains] != ref.flags.tolist() or ld.tops0 != ref.tops0:
                bad.append("ladder state")
                break
        return desc, bad
    # ptdc: class representatives as the seeds, one row of 16 / 4 per syndrome
    if name in ("xzzx", "rotated"):
        name, L = "toric", int(rng.choice([3, 4, 5]))
        ocode, qcode, shape = orc.TORIC, q.TORIC, (2, L, L)
        m = (rng.integers(1, 4, size=shape) * (rng.random(shape) < 0.15)).astype(np.uint8)
    ncls = 16 if name == "toric" else 4
    Nc, droplets, N = int(rng.choice([1, 1, 3, 4])), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    steps, iters = int(rng.integers(20, 300)), (5 if Nc == 1 else 10)
    per_rung = bool(Nc > 1 and rng.random() < 0.3)
    desc.update(code=name, L=L, Nc=Nc, droplets=droplets, N=N, steps=steps, per_rung=per_rung, p=min(p, 0.3))
    if name == "toric":
        reps = np.stack([q.toric_model.to_class(m, e) for e in range(16)])
    else:
        from qecmc import planar_model as pm
        reps = np.stack([pm.apply_logical(m, op, 0, 0)[0] for op in range(4)])
    init = np.stack([reps] * N)
    kw = dict(steps=steps, droplets=droplets, iters=iters, seed=seed & 0xFFFFFFFF, first_syndrome=stream, with_m=True, per_rung=per_rung)
    got_n, got_m = q.ptdc_batch(init, min(p, 0.3), Nc=Nc, code=qcode, **kw)
    ref_n, ref_m = orc.ptdc_batch(ocode, init, min(p, 0.3), Nc, kw.pop("steps"), **kw)
    bad = ([] if np.array_equal(got_n, ref_n) else ["N(n)"]) + ([] if np.array_equal(got_m, ref_m) else ["m(n)"])
    return desc, bad


def main_other(cases, seed):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    failures, kinds = [], {}
    for i in range(cases):
        try:
            desc, bad = run_other(rng)
        except q.QecmcError as e:
            kinds["refused"] = kinds.get("refused", 0) + 1
            print("refused:", e, flush=True)
            continue
        kinds[desc["kind"] + "/" + desc["code"]] = kinds.get(desc["kind"] + "/" + desc["code"], 0) + 1
        if bad:
            failures.append(dict(case=desc, differs=bad))
            print("MISMATCH", desc, bad, flush=True)
        if i % 50 == 49:
            print("%d cases, %d failures, %.0f s" % (i + 1, len(failures), time.time() - t0), flush=True)
    out = dict(mode="other", seed=seed, cases=cases, compared=sum(v for k, v in kinds.items() if k != "refused"), failures=failures, kinds=kinds,
               seconds=time.time() - t0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fuzz_other_%d.json" % seed), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "failures"}))
    print("FAILURES: %d" % len(failures))
    return 1 if failures else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "other":
        return main_other(int(sys.argv[2]) if len(sys.argv) > 2 else 300, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    failures, done, kinds = [], 0, {}
    for i in range(cases):
        c = draw_case(rng)
        if os.environ.get("QECMC_FUZZ_TRACE"):
            print("case %d %s" % (i, json.dumps(c)), flush=True)           # (what was running when a launch took the process down)
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
            k = "%s/%s/%s%s%s%s%s" % (c["code"], c["noise"], c["scan"], "/conv" if c["conv"] else "", "/queue-grid-" + c["grid"] if c["grid"] else "",
                                      "/chunked" if c["chunks"] else "", "/replicas" if c["R"] > 1 else "")
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
