#!/usr/bin/env python3
"""bench.py -- MCMC sweeps/s of the PTEQ hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: qecmc_pteq_launch_dev on
65 536 toric L=9 syndromes (p=0.15, Nc=8 temperatures, iters=10) for
`--ladder-steps` ladder steps, inputs already resident in HBM.  With N GPUs
every rank processes its own 65 536-syndrome shard (weak scaling; Philox keyed
by the global syndrome index) and the per-class counts are gathered to rank 0
over RCCL inside the timed region.

Prints ONE JSON line (rank 0).  `value` = chain-sweeps/s over all ranks, a sweep
being 2*L*L = 162 Metropolis proposals on one chain.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_PROPOSAL = 8  # 4 one-byte qubit reads + 4 writes (toric_model.py:275-278), SURVEY.md §8d


def synth_batch(N, L, p, seed):
    """Toric_code.generate_random_error(p) (toric_model.py:15-23) followed by one
    apply_random_logical (generate_data.py:131), vectorised over N syndromes."""
    rng = np.random.default_rng(seed)
    m = np.zeros((N, 2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    ops = rng.integers(0, 4, size=(N, 2))
    xpos = rng.integers(0, L, size=(N, 2))
    zpos = rng.integers(0, L, size=(N, 2))
    idx = np.arange(N)
    for layer in range(2):
        do_x = np.isin(ops[:, layer], (1, 2))
        do_z = np.isin(ops[:, layer], (3, 2))
        for i in range(L):
            if layer == 0:
                m[idx[do_x], 0, xpos[do_x, 0], i] ^= 1
                m[idx[do_z], 0, i, zpos[do_z, 0]] ^= 3
            else:
                m[idx[do_x], 1, i, xpos[do_x, 1]] ^= 1
                m[idx[do_z], 1, zpos[do_z, 1], i] ^= 3
    return m


def synth_batch_plaquette(N, L, px, py, pz, seed):
    """xzzx_code / RotSurCode.generate_random_error(p_x, p_y, p_z) (xzzx_model.py:16-30), vectorised;
    no logical is applied on top (the class of a raw error chain is what generate_data.py:121-122 records)."""
    rng = np.random.default_rng(seed)
    r = rng.random((N, L, L))
    m = np.zeros((N, L, L), dtype=np.uint8)
    m[r < pz] = 3
    m[(r > pz) & (r < pz + px)] = 1
    m[(r > pz + px) & (r < pz + px + py)] = 2
    return m


def measured_hbm_traffic(args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_summary.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes around this same default bench command),
    or None when the workload differs from the profiled one."""
    import glob
    if (args.scan, args.code, args.syndromes, args.L, args.Nc, args.iters, args.ladder_steps, args.p_logical) != ("random", "toric", 65536, 9, 8, 10, 2000, 0.5):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None
    # 1-byte-per-lane loads (64 B per wave instruction): no x2 FETCH_SIZE correction applies, see profiles/README.md
    return (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0


def cpu_baseline(init, p, Nc, iters, seed, target_s=12.0):
    """The oracle (CPU restatement with the reference's random-scan semantics) timed on this
    box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    L = init.shape[2]
    n_syn = min(init.shape[0], 4 * cores)
    orc.toric_pteq_batch(init[:n_syn], p, Nc, 200, iters=iters, tops_burn=2, seed=seed, n_threads=cores)   # spin the threads up
    t0 = time.perf_counter()
    orc.toric_pteq_batch(init[:n_syn], p, Nc, 2000, iters=iters, tops_burn=2, seed=seed, n_threads=cores)   # calibration
    dt = max(time.perf_counter() - t0, 1e-4)
    steps = int(max(2000, min(200000, 2000 * target_s / dt)))
    t0 = time.perf_counter()
    orc.toric_pteq_batch(init[:n_syn], p, Nc, steps, iters=iters, tops_burn=2, seed=seed, n_threads=cores)
    dt = time.perf_counter() - t0
    proposals = n_syn * Nc * iters * steps
    return {"value": proposals / (2 * L * L) / dt, "unit": "chain-sweeps/s", "cores": cores, "kind": "port",
            "sample": f"{n_syn} syndromes x {steps} ladder steps ({proposals:.3g} proposals, {dt:.1f} s, "
                      f"OpenMP over syndromes)",
            "proposals_per_s": proposals / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ladder-steps", type=int, default=2000, help="ladder steps per pass (decoders.py `steps`)")
    ap.add_argument("--syndromes", type=int, default=65536, help="syndromes per GPU")
    ap.add_argument("--L", type=int, default=9)
    ap.add_argument("--p", type=float, default=0.15)
    ap.add_argument("--Nc", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20200915)
    ap.add_argument("--p-logical", type=float, default=0.5, help="top-chain logical rate (decoders.py:52 uses 0.5)")
    ap.add_argument("--code", default="toric", choices=["toric", "xzzx", "rotated", "planar"], help="other codes: parity-test configs 4, 5")
    ap.add_argument("--eta", type=float, default=None, help="bias: selects the mcmc_biased chain (config 4)")
    ap.add_argument("--scan", default="random", choices=["random", "sweep"],
                    help="random = the reference's random-scan chain; sweep = systematic generator sweep (scan=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the extra scan=1 measurement (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qecmc import _lib as L_

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    N, L, Nc = args.syndromes, args.L, args.Nc
    toric = args.code == "toric"
    code_id = {"toric": L_.TORIC, "xzzx": L_.XZZX, "rotated": L_.ROTATED, "planar": L_.PLANAR}[args.code]
    nq, ncls = (2 * L * L, 16) if toric else (2 * L * L if args.code == "planar" else L * L, 4)
    first = rank * N                                    # global syndrome index of this shard
    if toric:
        init_h = synth_batch(N, L, args.p, args.seed + rank)
    elif args.code == "planar":       # Planar_code.generate_random_error(p/3, p/3, p/3) (generate_data.py:66-68, planar_model.py:18-40)
        init_h = np.stack([synth_batch_plaquette(N, L, args.p / 3, args.p / 3, args.p / 3, args.seed + rank + 7919 * l) for l in range(2)], axis=1)
        init_h[:, 1, -1, :] = 0
        init_h[:, 1, :, -1] = 0
    elif args.eta is None:
        init_h = synth_batch_plaquette(N, L, args.p / 3, args.p / 3, args.p / 3, args.seed + rank)
    else:
        init_h = synth_batch_plaquette(N, L, args.p / (2 * (args.eta + 1)), args.p / (2 * (args.eta + 1)),
                                       args.p * args.eta / (args.eta + 1), args.seed + rank)
    d_init = torch.from_numpy(init_h.reshape(N, nq)).to(dev)
    d_counts = torch.zeros((N, ncls), dtype=torch.int32, device=dev)
    d_samples = torch.zeros(N, dtype=torch.int32, device=dev)
    d_tops0 = torch.zeros(N, dtype=torch.int32, device=dev)
    gathered = [torch.zeros_like(d_counts) for _ in range(world)] if (use_dist and rank == 0) else None

    pr = L_.make_params(code=code_id, L=L, Nc=Nc, p=args.p, p_logical=args.p_logical, iters=args.iters,
                        steps=args.ladder_steps, tops_burn=2, seed=args.seed, device=local_rank,
                        noise=L_.NOISE_DEPOLARIZING if args.eta is None else L_.NOISE_BIASED, eta=args.eta or 0.0,
                        scan=L_.SCAN_RANDOM if args.scan == "random" else L_.SCAN_CHECKERBOARD)
    plan = C.c_void_p()
    L_.check(L_.lib().qecmc_plan_create(pr, C.byref(plan)))
    lds, threads, spb = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L_.check(L_.lib().qecmc_plan_info(plan, lds, threads, spb))
    stream = torch.cuda.current_stream()

    def one_pass():
        L_.check(L_.lib().qecmc_pteq_launch_dev(plan, d_init.data_ptr(), N, first, d_counts.data_ptr(),
                                                d_samples.data_ptr(), d_tops0.data_ptr(), None, None, None, None,
                                                C.c_void_p(stream.cuda_stream)))

    def exchange():
        if use_dist:                                    # the path's one exchange step: per-class counts -> rank 0
            dist.gather(d_counts, gathered, dst=0)

    for _ in range(args.warmup):
        one_pass(); exchange()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream); one_pass(); b.record(stream)
        exchange()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    proposals_per_pass = N * Nc * args.iters * args.ladder_steps      # per GPU
    total_proposals = proposals_per_pass * args.steps * world
    sweeps_per_s = total_proposals / nq / elapsed

    if rank == 0:
        samples = d_samples.cpu().numpy()
        tops0 = d_tops0.cpu().numpy()
        k_ms = float(np.mean(kernel_ms))
        algo_bytes = proposals_per_pass * ALGO_BYTES_PER_PROPOSAL + N * (nq + 4 * ncls)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "MCMC sweeps/sec (whole node), L=9 toric p=0.15; eq-class histogram match",
            "value": sweeps_per_s,
            "unit": "chain-sweeps/s (1 sweep = 2*L*L = %d Metropolis proposals on one chain)" % nq,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %s L=%d p=%g%s, %d syndromes per GPU, Nc=%d parallel tempering, "
                                   "iters=%d, %d ladder steps per pass, scan=%s"
                                   % ("configs[1]" if (toric and L == 9) else "parity-test configuration", args.code, L, args.p,
                                      "" if args.eta is None else " eta=%g" % args.eta, N, Nc, args.iters, args.ladder_steps,
                                      "random (the reference's chain)" if args.scan == "random" else "sweep (systematic generator sweep)"),
                       "syndromes_per_gpu": N, "L": L, "p": args.p, "Nc": Nc, "iters": args.iters,
                       "ladder_steps": args.ladder_steps, "tops_burn": 2, "seed": args.seed,
                       "lds_bytes_per_workgroup": lds.value, "threads_per_workgroup": threads.value,
                       "parallelism": "syndrome shards x%d, RCCL gather of class counts" % world},
            "proposals_per_s": total_proposals / elapsed,
            "ladder_sweeps_per_s": sweeps_per_s / Nc,
            "kernel_ms_per_launch": k_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_hbm_traffic(args),
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, profiles/)",
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "algorithmic bytes = 8 B/proposal + N*(nq+4*ncls) (SURVEY.md 8d); the state is "
                                 "LDS-resident so real HBM traffic is ~N*(nq+4*ncls+8) B per launch and the binding "
                                 "resource is instruction issue (Philox) with the LDS array close behind, see DESIGN.md"},
            "mixing": {"frac_syndromes_past_burn_in": float(np.mean(samples > 0)),
                       "mean_tops0": float(np.mean(tops0))},
        }
        if world == 1 and not args.no_cpu_baseline and toric and args.scan == "random":
            out["cpu_baseline"] = cpu_baseline(init_h, args.p, Nc, args.iters, args.seed)
            if args.p_logical == 0.5:
                # the metric's "eq-class histogram match": the oracle (the checker, on the same Philox streams) must give the
                # class counts the timed GPU pass left in HBM, bit for bit, on a sample of the batch
                from oracle import oracle as orc
                n_chk = min(N, 256)
                ref = orc.toric_pteq_batch(init_h[:n_chk], args.p, Nc, args.ladder_steps, iters=args.iters, tops_burn=2,
                                           seed=args.seed, n_threads=os.cpu_count() or 1)
                same = bool(np.array_equal(d_counts[:n_chk].cpu().numpy().astype(np.uint32), ref["counts"]) and
                            np.array_equal(samples[:n_chk].astype(np.uint64), ref["samples"].astype(np.uint64)))
                out["histogram_match"] = {"syndromes_checked": n_chk, "ladder_steps": args.ladder_steps,
                                          "class_counts_bit_identical_to_cpu_oracle": same}
        if world == 1 and toric and args.scan == "random" and args.eta is None and not args.no_sweep:
            # the library's second scan mode on the same batch, for the record (`value` above is the reference's chain)
            pr2 = L_.make_params(code=code_id, L=L, Nc=Nc, p=args.p, p_logical=args.p_logical, iters=args.iters,
                                 steps=args.ladder_steps, tops_burn=2, seed=args.seed, device=local_rank,
                                 scan=L_.SCAN_CHECKERBOARD)
            plan2 = C.c_void_p()
            L_.check(L_.lib().qecmc_plan_create(pr2, C.byref(plan2)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(2):
                e0.record(stream)
                L_.check(L_.lib().qecmc_pteq_launch_dev(plan2, d_init.data_ptr(), N, first, d_counts.data_ptr(),
                                                        d_samples.data_ptr(), d_tops0.data_ptr(), None, None, None, None,
                                                        C.c_void_p(stream.cuda_stream)))
                e1.record(stream)
                torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1)
            out["sweep_scan"] = {"note": "scan=1: systematic generator sweep (not the reference's chain; same stationary law, "
                                         "validated against exact enumeration), same batch and ladder steps",
                                 "kernel_ms_per_launch": ms2, "proposals_per_s": proposals_per_pass / (ms2 * 1e-3),
                                 "chain_sweeps_per_s": proposals_per_pass / nq / (ms2 * 1e-3)}
            L_.lib().qecmc_plan_destroy(plan2)
        print(json.dumps(out))
    L_.lib().qecmc_plan_destroy(plan)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
