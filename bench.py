#!/usr/bin/env python3
"""bench.py -- MCMC sweeps/s of the PTEQ hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: qecmc_pteq_launch_dev on the syndromes of one of BASELINE.json's
configurations (`--config`, default 2 = the headline: 65 536 toric L=9 syndromes, p=0.15, Nc=8 temperatures, iters=10) for
`--ladder-steps` ladder steps, inputs already resident in HBM.  With N GPUs every rank processes its own shard of the same
size (weak scaling; Philox keyed by the global syndrome index) and the per-class counts are gathered to rank 0 over RCCL
inside the timed region (qecmc.sharding.PteqShard: the library's own sharded call is what is timed).

    python bench.py                      1 GPU, config 2
    python bench.py --gpus 8             starts 8 ranks itself (python -m torch.distributed.run ... bench.py --gpus 8)
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8      the driver's form: ranks already exist
    python bench.py --config 3|4|5       the other BASELINE configurations (one line each, same fields)

Prints ONE JSON line (rank 0).  `value` = chain-sweeps/s over all ranks, a sweep being G Metropolis proposals on one chain,
G = number of stabilizer generators (2 L^2 toric, L^2 - 1 xzzx / rotated).
"""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mcmc-qec-toric-rl_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_PROPOSAL = 8  # 4 one-byte qubit reads + 4 writes (toric_model.py:275-278), SURVEY.md §8d

# BASELINE.json `configs` (index = position in the list, 1-based; configs[0] is the reference's own CPU case)
CONFIGS = {
    2: dict(code="toric", L=9, p=0.15, Nc=8, syndromes=65536, eta=None, name="configs[1]: toric L=9 p=0.15, 65 536 syndromes, 8-temperature PT"),
    3: dict(code="toric", L=15, p=0.18, Nc=8, syndromes=131072, eta=None, name="configs[2]: toric L=15 p=0.18, 1M syndromes over 8 GPUs = 131 072 per GPU"),
    4: dict(code="xzzx", L=9, p=0.15, Nc=8, syndromes=65536, eta=100.0, name="configs[3]: XZZX L=9 biased noise eta=100 (mcmc_biased chain)"),
    5: dict(code="rotated", L=21, p=0.17, Nc=8, syndromes=32768, eta=None, name="configs[4]: rotated L=21 p=0.17 long-chain study shape, 32 768 syndromes per GPU (2 workgroups per CU)"),
}


def synth_batch(N, L, p, seed, return_raw=False):
    """Toric_code.generate_random_error(p) (toric_model.py:15-23) followed by one
    apply_random_logical (generate_data.py:131), vectorised over N syndromes.  return_raw: also the error chains before the
    logical operator (their class is the decoding target, generate_data.py:121-122,139)."""
    rng = np.random.default_rng(seed)
    m = np.zeros((N, 2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    raw = m.copy() if return_raw else None
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
    return (m, raw) if return_raw else m


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


def make_batch(args, rank):
    N, L = args.syndromes, args.L
    if args.code == "toric":
        return synth_batch(N, L, args.p, args.seed + rank)
    if args.code == "planar":   # Planar_code.generate_random_error(p/3, p/3, p/3) (generate_data.py:66-68, planar_model.py:18-40)
        m = np.stack([synth_batch_plaquette(N, L, args.p / 3, args.p / 3, args.p / 3, args.seed + rank + 7919 * l) for l in range(2)], axis=1)
        m[:, 1, -1, :] = 0
        m[:, 1, :, -1] = 0
        return m
    if args.eta is None:
        return synth_batch_plaquette(N, L, args.p / 3, args.p / 3, args.p / 3, args.seed + rank)
    return synth_batch_plaquette(N, L, args.p / (2 * (args.eta + 1)), args.p / (2 * (args.eta + 1)),
                                 args.p * args.eta / (args.eta + 1), args.seed + rank)    # generate_data.py:78-83


def profile_counters(args):
    """What the committed rocprofv3 passes of this exact workload say binds the kernel (profiles/rNN_cfgC_pmc_summary.json,
    written by tools/profile_round.sh around this same command): HBM bytes per launch and the issue / LDS utilisation.
    None when the workload differs from the profiled one."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", ("r*_alpha_pmc_summary.json" if getattr(args, "alpha_route", False) else "r*_cfg%d_pmc_summary.json" % args.config))))
    if not files:
        return None
    d = json.load(open(files[-1]))
    w = d.get("_workload", {})
    same = (all(w.get(k) == getattr(args, k) for k in ("code", "L", "Nc", "iters", "syndromes", "ladder_steps")) and w.get("p_logical", 0.5) == args.p_logical
            and w.get("scan", "random") == args.scan and bool(w.get("alpha_route", False)) == bool(getattr(args, "alpha_route", False)))
    if not same or "SQ_INSTS_VALU" not in d:
        return None
    wave_props = args.syndromes / 64 * args.Nc * args.iters * args.ladder_steps         # wave-proposals per launch
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0                                              # per-XCD active cycles of the launch
    out = {"source": os.path.basename(files[-1]),
           # gfx950 counts FETCH_SIZE in units of 64 B where the counter's documentation says 32 B (MI355X_MICROARCH.md, HBM / rocprofv3
           # section): the raw value times 2; WRITE_SIZE as reported.  (Round 3 printed the uncorrected sum, which sat below the compulsory bytes.)
           "traffic": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in d and "WRITE_SIZE" in d else None,
           "insts_per_wave_proposal": {"valu": d["SQ_INSTS_VALU"] / wave_props, "salu": d["SQ_INSTS_SALU"] / wave_props,
                                       "lds": d["SQ_INSTS_LDS"] / wave_props}}
    if cyc and "SQ_ACTIVE_INST_VALU" in d:
        simd_cycles = cyc * 256 * 4              # SIMD-cycles of the launch: 256 CUs x 4 SIMDs
        out["valu_busy"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / simd_cycles   # the SQ_ACTIVE_* / SQ_WAVE_CYCLES counters tick in quad-cycles (MI355X_MICROARCH.md)
        out["lds_busy"] = d["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)       # one LDS array per CU
        out["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1.0)
    return out


def metric_name(args):
    """BASELINE.json's metric, named after the workload that was run (config 2 gives BASELINE's own string)"""
    return "MCMC sweeps/sec (whole node), L=%d %s p=%g%s%s; eq-class histogram match" % (
        args.L, args.code, args.p, "" if args.eta is None else " eta=%g" % args.eta,
        " decoded by PTEQ_alpha(pz_tilde=%.4g, alpha=%.4g)" % rule(args) if getattr(args, "alpha_route", False) else "")


def rule(args):
    """What the decoder is handed: (p, keyword arguments of the rule) for the library (`lib=True`: noise ids of qecmc._lib) and the oracle.
    --alpha-route: errors drawn at (p, eta), decoded by PTEQ_alpha(pz_tilde, alpha) -- generate_data.py:142-150's route for biased noise."""
    if getattr(args, "alpha_route", False):
        pz_tilde = (args.p / (1 + 1 / args.eta)) / (1 - args.p)                          # generate_data.py:145
        return float(pz_tilde), float(np.log(pz_tilde / (2 * args.eta)) / np.log(pz_tilde))   # :146
    return args.p, None


def lib_rule(args, L_):
    p_dec, a = rule(args)
    if a is not None:
        return p_dec, dict(noise=L_.NOISE_ALPHA, eta=0.0, alpha=a)
    return p_dec, dict(noise=L_.NOISE_DEPOLARIZING if args.eta is None else L_.NOISE_BIASED, eta=args.eta or 0.0)


def orc_rule(args, orc):
    p_dec, a = rule(args)
    if a is not None:
        return p_dec, dict(noise=orc.ALPHA, alpha=a, det_pow=1)
    return p_dec, dict(noise=orc.DEPOLARIZING if args.eta is None else orc.BIASED, eta=args.eta or 0.0)


def api_rule(args):
    p_dec, a = rule(args)
    return p_dec, (dict(alpha=a) if a is not None else dict(eta=args.eta))


def rank_records(dist, use_dist, world, rank, device_name, device_uuid, kernel_ms, first):
    """What every rank ran, gathered to rank 0 so that an N > 1 line certifies itself: the ranks RCCL saw, the device each one
    sat on, its own kernel time and the first global syndrome index of its shard."""
    mine = {"rank": rank, "device": device_name, "device_uuid": device_uuid, "kernel_ms_mean": kernel_ms, "first_syndrome": first}
    if not use_dist:
        return [mine]
    got = [None] * world
    dist.all_gather_object(got, mine)
    return got


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_batch(args, init, steps, n_threads, first=0, states=False, scan=None, tops_burn=2):
    """the CPU oracle on a sample of the batch: the checker of the timed pass (its scan) or the timed CPU baseline (scan = 0: the
    reference's own loop, one pick per ladder -- a CPU has no wavefront to share a pick with)"""
    from oracle import oracle as orc
    code = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED, "planar": orc.PLANAR}[args.code]
    p_dec, okw = orc_rule(args, orc)
    return orc.pteq_batch(code, init, p_dec, args.Nc, steps, iters=args.iters, tops_burn=tops_burn, seed=args.seed, first_syndrome=first,
                          n_threads=n_threads, return_states=states, scan=(3 if args.scan == "wave" else 0) if scan is None else scan, **okw)


def cpu_baseline(args, init, n_gen, target_s=12.0):
    """The oracle (CPU restatement with the reference's random-scan semantics) timed on this
    box's host cores on a bounded sample of the same workload."""
    cores = os.cpu_count() or 1
    n_syn = min(init.shape[0], 4 * cores)
    oracle_batch(args, init[:n_syn], 100, cores, scan=0)            # spin the threads up
    t0 = time.perf_counter()
    oracle_batch(args, init[:n_syn], 500, cores, scan=0)            # calibration
    dt = max(time.perf_counter() - t0, 1e-4)
    steps = int(max(500, min(200000, 500 * target_s / dt)))
    t0 = time.perf_counter()
    oracle_batch(args, init[:n_syn], steps, cores, scan=0)
    dt = time.perf_counter() - t0
    proposals = n_syn * args.Nc * args.iters * steps
    return {"value": proposals / n_gen / dt, "unit": "chain-sweeps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{n_syn} syndromes x {steps} ladder steps ({proposals:.3g} proposals, {dt:.1f} s, "
                      f"OpenMP over syndromes, {cores} threads)",
            "proposals_per_s": proposals / dt}


def criterion_bench(args, L_, PteqShard, torch, dist, use_dist, world, rank, local_rank, dev, init_h, first, code_id, nq, ncls, n_gen, workload):
    """--criterion: the reference's default route (decoders.py:25,74-105).  One step = one launch of the batch through the criterion
    kernels on their persistent grid; `value` = chain-sweeps the ladders actually ran / wall time (a ladder that stops early stops
    counting); roofline on the same 8 B / proposal convention, useful proposals only."""
    N, L, Nc = args.syndromes, args.L, args.Nc
    crit = dict(conv_mode=L_.CONV_ERROR_BASED, SEQ=2, TOPS=10, eps=0.1)                 # decoders.py:25
    p_dec, rkw = lib_rule(args, L_)
    common = dict(code=code_id, Nc=Nc, p_logical=args.p_logical, iters=args.iters, tops_burn=2, seed=args.seed, scan=L_.SCANS[args.scan], flags=args.flags, **rkw)
    sh = PteqShard(init_h, p_dec, first, n_total=N * world, steps=args.ladder_steps, **common, **crit)
    lds, threads, spb = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L_.check(L_.lib().qecmc_plan_info(sh.plan, lds, threads, spb))
    stream = torch.cuda.current_stream()
    for _ in range(args.warmup):
        sh.launch(stream); sh.gather()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream); sh.launch(stream); b.record(stream)
        sh.gather()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    useful_steps = torch.tensor([float(sh.steps_done.to(torch.float64).sum().item())], dtype=torch.float64, device=dev)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(useful_steps, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    props = torch.cuda.get_device_properties(dev)
    ranks = rank_records(dist, use_dist, world, rank, props.name, str(getattr(props, "uuid", "")) or None, float(np.mean(kernel_ms)), first)
    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        sd = sh.steps_done.cpu().numpy().astype(np.float64)
        conv = sh.converged.cpu().numpy().astype(bool)
        useful_props_launch = float(sd.sum()) * Nc * args.iters                          # this rank's launch
        useful_total = float(useful_steps.item()) * Nc * args.iters * args.steps          # all ranks, all timed launches
        algo_bytes = useful_props_launch * ALGO_BYTES_PER_PROPOSAL + N * (nq + 4 * ncls)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        # the same kernel family at fixed length on the first 65 536 syndromes: what a ladder step costs without the criterion, the queue
        # and its tail (time after a workgroup's queue ran dry)
        n_fix = min(N, 65536)
        shf = PteqShard(init_h[:n_fix], p_dec, first, n_total=n_fix, steps=10000, **common)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fix_ms = []
        for rep in range(3):
            e0.record(stream); shf.launch(stream); e1.record(stream)
            torch.cuda.synchronize()
            fix_ms.append(e0.elapsed_time(e1))
        shf.close()
        fixed_rate = n_fix * Nc * args.iters * 10000 / (float(np.mean(fix_ms[1:])) * 1e-3)
        out = {
            "metric": metric_name(args) + " -- criterion-stopped (decoders.py:25 conv_criteria='error_based', SEQ=2, TOPS=10, eps=0.1)",
            "value": useful_total / n_gen / elapsed,
            "unit": "chain-sweeps/s of the steps the ladders ran (1 sweep = %d Metropolis proposals on one chain)" % n_gen,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload + "; every ladder stopped by the error_based criterion or at the horizon of %d of its own steps" % args.ladder_steps,
                       "baseline_config": args.config, "syndromes_per_gpu": N, "code": args.code, "L": L, "p": args.p, "eta": args.eta, "Nc": Nc,
                       "decoder": ("PTEQ_alpha(pz_tilde=%.6g, alpha=%.6g), generate_data.py:142-150" % rule(args)) if args.alpha_route else "PTEQ_biased" if args.eta is not None else "PTEQ",
                       "iters": args.iters, "horizon_ladder_steps": args.ladder_steps, "SEQ": 2, "TOPS": 10, "eps": 0.1, "tops_burn": 2, "seed": args.seed,
                       "scan": args.scan, "lds_bytes_per_workgroup": lds.value, "threads_per_workgroup": threads.value,
                       "workspace_bytes": sh.ws_bytes, "parallelism": "syndrome shards x%d, RCCL gather of class counts" % world},
            "useful_proposals_per_launch": useful_props_launch,
            "useful_proposals_per_s": useful_total / elapsed,
            "kernel_ms_per_launch": k_ms,
            "stopping": {"converged_frac": float(conv.mean()), "mean_steps": float(sd.mean()), "median_steps": float(np.median(sd)),
                         "p99_steps": float(np.percentile(sd, 99)), "max_steps": float(sd.max())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "8 B per USEFUL proposal (SURVEY.md 8d convention): steps a ladder ran before it stopped; idle lanes, the two "
                                 "steps a lane waits between ladders and the tail after a workgroup's queue ran dry are time, not work"},
            # useful rate / the fixed-length rate of the same kernel family on this box: what the criterion, the queue and its tail cost together
            "efficiency_vs_fixed_length": (useful_props_launch / (k_ms * 1e-3)) / fixed_rate,
            "fixed_length_proposals_per_s": fixed_rate,
            "comm": {"backend": (dist.get_backend() + " (RCCL)") if use_dist else None, "world": dist.get_world_size() if use_dist else 1},
            "ranks": ranks,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as orc
            ocode = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED, "planar": orc.PLANAR}[args.code]
            cores = os.cpu_count() or 1
            n_cpu = min(N, cores)
            okw = dict(iters=args.iters, tops_burn=2, seed=args.seed, n_threads=cores, conv_criteria="error_based", SEQ=2, TOPS=10, eps=0.1, **orc_rule(args, orc)[1])
            t1 = time.perf_counter()
            ref0 = orc.pteq_batch(ocode, init_h[:n_cpu], p_dec, Nc, args.ladder_steps, first_syndrome=first, scan=0, **okw)   # the reference's own loop
            dt = time.perf_counter() - t1
            cpu_props = float(ref0["steps_done"].sum()) * Nc * args.iters
            out["cpu_baseline"] = {"value": cpu_props / n_gen / dt, "unit": "chain-sweeps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
                                   "sample": f"{n_cpu} syndromes, one per thread, each run to its criterion stop or the horizon ({cpu_props:.3g} proposals, {dt:.1f} s)",
                                   "proposals_per_s": cpu_props / dt}
            # the metric's histogram match on the criterion route: 64 ladders (one workgroup, a lane each) through the same entry point and
            # kernel, against the oracle's restatement of the rule -- class counts, samples, tops0, stopping step and flag, bit for bit
            import qecmc
            n_chk = min(N, 64)
            got = qecmc.pteq_batch(init_h[:n_chk], p_dec, Nc=Nc, steps=args.ladder_steps, iters=args.iters, tops_burn=2, seed=args.seed, first_syndrome=first,
                                   code=code_id, conv_criteria="error_based", SEQ=2, TOPS=10, eps=0.1, scan=args.scan, flags=args.flags, **api_rule(args)[1])
            ref = orc.pteq_batch(ocode, init_h[:n_chk], p_dec, Nc, args.ladder_steps, first_syndrome=first, scan=3 if args.scan == "wave" else 0, **okw)
            same = all(np.array_equal(np.asarray(got[k]).astype(np.uint64), np.asarray(ref[k]).astype(np.uint64)) for k in ("counts", "samples", "tops0", "steps_done"))
            out["histogram_match"] = {"syndromes_checked": n_chk, "horizon": args.ladder_steps,
                                      "class_counts_samples_tops0_stopping_step_bit_identical_to_cpu_oracle": bool(same),
                                      "converged_flags_identical": bool(np.array_equal(got["converged"], ref["converged"])),
                                      "match": bool(same and np.array_equal(got["converged"], ref["converged"]))}
        print(json.dumps(out))
    sh.close()
    if use_dist:
        dist.destroy_process_group()


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside a launcher: start N ranks (one per GPU) as children of this process, which has not
    touched the GPU, wait, and hand their exit code on.  The same command line the driver uses."""
    from qecmc.sharding import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (2 = the metric's)")
    ap.add_argument("--ladder-steps", type=int, default=10000, help="ladder steps per pass (decoders.py `steps`; SURVEY.md 8d: 1e4 for throughput)")
    ap.add_argument("--syndromes", type=int, default=None, help="syndromes per GPU")
    ap.add_argument("--L", type=int, default=None)
    ap.add_argument("--p", type=float, default=None)
    ap.add_argument("--Nc", type=int, default=None)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20200915)
    ap.add_argument("--p-logical", type=float, default=0.5, help="top-chain logical rate (decoders.py:52 uses 0.5)")
    ap.add_argument("--code", default=None, choices=["toric", "xzzx", "rotated", "planar"])
    ap.add_argument("--eta", type=float, default=None, help="bias: selects the mcmc_biased chain (config 4)")
    ap.add_argument("--alpha-route", action="store_true",
                    help="the route generate_data.py:142-150 takes for biased noise: errors drawn at (p, eta) on the xzzx code (defaults L=5, p=0.15, "
                         "eta=100, Nc=5), decoded by PTEQ_alpha with (pz_tilde, alpha) derived from them (src/mcmc_alpha.py)")
    ap.add_argument("--scan", default="auto", choices=["auto", "random", "sweep", "wave"],
                    help="random = the reference's random-scan chain (scan=0); wave = the same chain per syndrome with a generator pick shared by the 64 "
                         "ladders of a wavefront, states in registers (scan=3); auto = wave where it is built and measured faster (toric code, depolarizing "
                         "rule, L <= 11 -- fixed-length runs L <= 16 --, and the alpha route), else random; sweep = systematic generator sweep (scan=1: not the reference's chain)")
    ap.add_argument("--criterion", action="store_true",
                    help="time the route the reference runs by default (decoders.py:25 conv_criteria='error_based', SEQ=2, TOPS=10, eps=0.1): every "
                         "ladder stops by the criterion or at the horizon of --ladder-steps of its own steps; `value` counts the steps the ladders "
                         "actually ran.  Defaults: 8 ladders per lane of the persistent grid (--syndromes 524288), horizon 262144, one launch per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--library", default=None, help="time another build of libqecmc.so (same ABI) instead of the in-tree one: A/B runs on one box")
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0, help="qecmc_params.flags: developer switches between equivalent kernel variants (include/qecmc.h)")
    ap.add_argument("--sweep", action="store_true", help="also time the scan=1 kernel on the same batch (toric, 1 GPU)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / exchange rehearsal without a GPU: gloo, no kernel, zero throughput (tests/test_bench_launcher.py)")
    args = ap.parse_args(argv)
    cfg = dict(CONFIGS[args.config])
    if args.alpha_route:
        cfg.update(name="generate_data.py:142-150: biased noise decoded by PTEQ_alpha", code="xzzx", L=5, p=0.15, eta=100.0, Nc=5, syndromes=98304)
        if args.criterion and args.ladder_steps == 10000:
            args.ladder_steps = 65536
    if args.criterion:
        # N >= 8 ladders per lane of the persistent grid (256 CUs x 4 workgroups x 64 lanes at the config-2 shape); the horizon bounds the
        # log workspace (2 B per lane of the grid and ladder step) and censors a quarter of the L = 9, p = 0.15 ladders
        if args.syndromes is None:
            # (the alpha route's ladders are short and heavy-tailed -- mean 13 000 steps, horizon 65 536 --: 16 per lane keep the tail of a
            # workgroup's queue, the time its last ladders run alone, to a quarter of the launch)
            args.syndromes = (16 if args.alpha_route else 8) * cfg["syndromes"]
        if args.ladder_steps == 10000:
            args.ladder_steps = 262144
        if args.steps == 5 and args.warmup == 1:      # (a launch runs for seconds: two timed ones, the first launch of the process among them)
            args.steps, args.warmup = 2, 0
    for k in ("code", "L", "p", "Nc", "syndromes", "eta"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    if args.scan == "auto":
        nq = (2 if args.code in ("toric", "planar") else 1) * args.L * args.L
        # scan = wave where same-box A/B runs have it ahead (profiles/r04_wave_ab.json): rung states of at most 16 words keep 8 waves per SIMD
        W = (nq + 15) // 16
        # where same-box A/B runs have scan = wave ahead (profiles/r04_wave_ab.json): up to 16 state words per rung with the unrolled loop of iters = 10
        # (L = 9: toric 0.81, xzzx / rotated / planar 0.82-0.85 against 0.66-0.73; the general loop of other iters: 0.51-0.61, behind the random scan) or on
        # more than 8 rungs (toric L = 9, Nc = 9 / 12 / 16: 0.71 / 0.76 / 0.79 against 0.31 / 0.38 / 0.43), the toric code's fixed-length runs up to 32 words
        # (config 3), the alpha rule; not rotated L = 21 at BASELINE's 32 768 syndromes (0.46 against 0.51)
        ok = W <= 8 and args.code in ("xzzx", "rotated") if args.alpha_route else (
            args.eta is None and (W <= 16 and (args.iters == 10 or args.Nc > 8) or
                                  (args.code == "toric" and 16 < W <= 32 and args.Nc <= 8 and args.iters == 10 and not args.criterion)))
        args.scan = "wave" if (ok and args.Nc >= 2 and args.iters <= 128 and args.syndromes % 64 == 0) else "random"
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))               # before any GPU / torch.cuda call in this process

    import torch
    import torch.distributed as dist
    from qecmc import _lib as L_
    from qecmc.sharding import PteqShard
    if args.library:
        L_.use_library(args.library)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    N, L, Nc = args.syndromes, args.L, args.Nc
    toric = args.code == "toric"
    code_id = {"toric": L_.TORIC, "xzzx": L_.XZZX, "rotated": L_.ROTATED, "planar": L_.PLANAR}[args.code]
    nq, ncls = (2 * L * L, 16) if toric else (2 * L * L if args.code == "planar" else L * L, 4)
    n_gen = 2 * L * L if toric else 2 * L * (L - 1) if args.code == "planar" else L * L - 1     # proposals per sweep
    first = rank * N                                    # global syndrome index of this shard
    proposals_per_pass = N * Nc * args.iters * args.ladder_steps      # per GPU
    workload = ("%s; %s L=%d p=%g%s, %d syndromes per GPU, Nc=%d parallel tempering, iters=%d, %d ladder steps per pass, scan=%s"
                % ("generate_data.py:142-150: biased noise decoded by PTEQ_alpha (pz_tilde=%.4g, alpha=%.4g)" % rule(args) if args.alpha_route else CONFIGS[args.config]["name"], args.code, L, args.p, "" if args.eta is None else " eta=%g" % args.eta, N, Nc,
                   args.iters, args.ladder_steps, {"random": "random (the reference's chain)", "wave": "wave (the reference's chain per syndrome; one generator pick per wavefront)", "sweep": "sweep"}[args.scan]))

    if args.dry_run:
        # launcher + exchange rehearsal (CPU, gloo): every rank fills its records with a rank-dependent pattern, rank 0 checks
        # the gather; no kernel runs and no throughput is claimed
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_dist:
            dist.init_process_group("gloo")
        rec = torch.full((N * (ncls + 2),), rank + 1, dtype=torch.int32)
        gathered = [torch.empty_like(rec) for _ in range(world)] if (use_dist and rank == 0) else None
        t0 = time.perf_counter()
        for _ in range(args.warmup + args.steps):
            if use_dist:
                dist.gather(rec, gathered, dst=0)
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        ranks = rank_records(dist, use_dist, world, rank, "cpu (dry run)", None, 0.0, first)
        if rank == 0:
            ok = all(int(g[0]) == r + 1 and int(g[-1]) == r + 1 for r, g in enumerate(gathered)) if use_dist else True
            print(json.dumps({"metric": "dry run: launcher and exchange only (no kernel, no throughput)", "value": 0.0, "unit": "chain-sweeps/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                              "data": "none (dry run)", "gather_ok": bool(ok), "config": {"workload": "dry run of: " + workload},
                              "comm": {"backend": dist.get_backend() if use_dist else None, "world": dist.get_world_size() if use_dist else 1},
                              "ranks": ranks}))
        if use_dist:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the product has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    init_h = make_batch(args, rank)
    if args.criterion:
        return criterion_bench(args, L_, PteqShard, torch, dist, use_dist, world, rank, local_rank, dev, init_h, first, code_id, nq, ncls, n_gen, workload)
    p_dec, rkw = lib_rule(args, L_)
    sh = PteqShard(init_h, p_dec, first, n_total=N * world, code=code_id, Nc=Nc, p_logical=args.p_logical, iters=args.iters,
                   steps=args.ladder_steps, tops_burn=2, seed=args.seed, scan=L_.SCANS[args.scan], flags=args.flags, **rkw)
    lds, threads, spb = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L_.check(L_.lib().qecmc_plan_info(sh.plan, lds, threads, spb))
    stream = torch.cuda.current_stream()

    for _ in range(args.warmup):
        sh.launch(stream); sh.gather()
    if use_dist and args.warmup == 0:
        sh.gather()                                             # (untimed: RCCL sets up its point-to-point channels at the first exchange)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream); sh.launch(stream); b.record(stream)   # HIP events on the stream the kernel is launched on
        sh.gather()                                             # the path's one exchange step: per-class counts -> rank 0
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

    total_proposals = proposals_per_pass * args.steps * world
    sweeps_per_s = total_proposals / n_gen / elapsed
    props = torch.cuda.get_device_properties(dev)
    ranks = rank_records(dist, use_dist, world, rank, props.name, str(getattr(props, "uuid", "")) or None, float(np.mean(kernel_ms)), first)

    if rank == 0:
        d_counts, d_samples, d_tops0 = sh.views()
        samples = d_samples.cpu().numpy()
        tops0 = d_tops0.cpu().numpy()
        k_ms = float(np.mean(kernel_ms))
        algo_bytes = proposals_per_pass * ALGO_BYTES_PER_PROPOSAL + N * (nq + 4 * ncls)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        pc = profile_counters(args) or {}
        # the host-pointer boundary (what a caller holding NumPy arrays pays): H2D of the batch + kernel + D2H of the records
        pinned = torch.from_numpy(init_h.reshape(N, -1)).pin_memory()
        rec_h = torch.empty_like(sh.rec, device="cpu").pin_memory()
        incl = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sh.d_init.copy_(pinned, non_blocking=True)
            sh.launch(stream)
            rec_h.copy_(sh.rec, non_blocking=True)
            torch.cuda.synchronize()
            incl.append((time.perf_counter() - t1) * 1e3)
        out = {
            "metric": metric_name(args),
            "value": sweeps_per_s,
            "unit": "chain-sweeps/s (1 sweep = %d Metropolis proposals on one chain = one per stabilizer generator)" % n_gen,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.config,
                       "syndromes_per_gpu": N, "code": args.code, "L": L, "p": args.p, "eta": args.eta, "Nc": Nc, "iters": args.iters,
                       "ladder_steps": args.ladder_steps, "tops_burn": 2, "seed": args.seed, "scan": args.scan,
                       "lds_bytes_per_workgroup": lds.value, "threads_per_workgroup": threads.value,
                       "parallelism": "syndrome shards x%d, RCCL gather of class counts" % world},
            "proposals_per_s": total_proposals / elapsed,
            "library_path": L_.library_path(),
            "comm": {"backend": (dist.get_backend() + " (RCCL)") if use_dist else None, "world": dist.get_world_size() if use_dist else 1},
            "ranks": ranks,
            "ladder_sweeps_per_s": sweeps_per_s / Nc,
            "kernel_ms_per_launch": k_ms,
            "ms_per_step_incl_transfers": float(np.median(incl)),
            "transfers_note": "pinned H2D of the %.1f MB batch + kernel + D2H of the %.1f MB records, 1 GPU, median of 3 (never `value`)"
                              % (init_h.nbytes / 1e6, sh.rec.numel() * 4 / 1e6),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pc.get("traffic"),
                         "traffic_unit": "bytes per launch (rocprofv3 2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md; profiles/)",
                         "compulsory_bytes_per_launch": N * (nq + 4 * ncls + 8),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         # what actually binds (instruction issue and the LDS array), from the committed PMC passes of this workload
                         "counters": {k: v for k, v in pc.items() if k != "traffic"} or None,
                         "note": "algorithmic bytes = 8 B/proposal + N*(nq+4*ncls) (SURVEY.md 8d); the states are on-chip (registers / LDS), "
                                 "so real HBM traffic is the compulsory N*(nq+4*ncls+8) B per launch and the binding resource is instruction "
                                 "issue, see DESIGN.md"},
            "mixing": {"frac_syndromes_past_burn_in": float(np.mean(samples > 0)),
                       "mean_tops0": float(np.mean(tops0)), "frac_tops0_ge_10": float(np.mean(tops0 >= 10))},
        }
        if world == 1 and not args.no_cpu_baseline and args.scan in ("random", "wave"):
            out["cpu_baseline"] = cpu_baseline(args, init_h, n_gen)
            if args.p_logical == 0.5:
                # the metric's "eq-class histogram match": the oracle (the checker, on the same Philox streams) must give the
                # class counts the timed GPU pass left in HBM, bit for bit, on a sample of the batch (the statistical
                # match against the reference's own sampler is tests/test_gpu_stats.py and profiles/rNN_headline_S1e5.json)
                n_chk = min(N, 256 if args.ladder_steps <= 20000 else 64)
                ref = oracle_batch(args, init_h[:n_chk], args.ladder_steps, os.cpu_count() or 1, states=True)
                with_samples = int(np.sum(samples[:n_chk] > 0))
                same = bool(np.array_equal(d_counts[:n_chk].cpu().numpy().astype(np.uint32), ref["counts"]) and
                            np.array_equal(samples[:n_chk].astype(np.uint64), ref["samples"].astype(np.uint64)))
                # state-dependent outputs that are non-zero whatever the burn-in did (the long-lattice configurations pass
                # tops_burn = 2 for few or none of their ladders in 10 000 steps): tops0 of the timed pass, and -- from one more
                # pass of the same ladders through the host-pointer entry point -- every rung's final configuration
                # That second pass runs with tops_burn = 0 (every step is a sample), so that the class histogram itself is compared on
                # every configuration: same trajectories, non-zero counts whatever the burn-in did.
                import qecmc
                again = qecmc.pteq_batch(init_h[:n_chk], p_dec, Nc=Nc, steps=args.ladder_steps, iters=args.iters, tops_burn=0, seed=args.seed,
                                         first_syndrome=first, code=code_id, return_states=True, flags=args.flags, scan=args.scan, **api_rule(args)[1])
                ref0 = oracle_batch(args, init_h[:n_chk], args.ladder_steps, os.cpu_count() or 1, tops_burn=0)
                hist0 = bool(np.array_equal(again["counts"], ref0["counts"]) and int(again["counts"].sum()) == n_chk * args.ladder_steps)
                out["histogram_match"] = {"syndromes_checked": n_chk, "ladder_steps": args.ladder_steps,
                                          "syndromes_with_samples": with_samples,
                                          # (null: no ladder of the sample got past the burn-in, the counts are all zero on both sides)
                                          "class_counts_bit_identical_to_cpu_oracle": same if with_samples else None,
                                          "class_histogram_of_every_step_bit_identical_to_cpu_oracle": hist0,
                                          "tops0_bit_identical_to_cpu_oracle": bool(np.array_equal(tops0[:n_chk].astype(np.uint64), ref["tops0"].astype(np.uint64))),
                                          "final_states_bit_identical_to_cpu_oracle": bool(np.array_equal(again["states"], ref["states"])),
                                          "match": bool(same and hist0 and np.array_equal(tops0[:n_chk].astype(np.uint64), ref["tops0"].astype(np.uint64)) and
                                                        np.array_equal(again["states"], ref["states"]))}
        if world == 1 and args.scan == "wave":
            # the same batch through the scan = 0 kernel (the chain pinned draw for draw to the reference's injected-stream fixtures), for the record
            pr0 = L_.make_params(code=code_id, L=L, Nc=Nc, p=p_dec, p_logical=args.p_logical, iters=args.iters, steps=args.ladder_steps, tops_burn=2,
                                 seed=args.seed, device=local_rank, scan=L_.SCAN_RANDOM, flags=args.flags, **rkw)
            plan0 = C.c_void_p()
            L_.check(L_.lib().qecmc_plan_create(pr0, C.byref(plan0)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms0 = []
            for rep in range(3):
                e0.record(stream)
                L_.check(L_.lib().qecmc_pteq_launch_dev(plan0, sh.d_init.data_ptr(), N, first, d_counts.data_ptr(), d_samples.data_ptr(), d_tops0.data_ptr(),
                                                        None, None, None, None, 0, C.c_void_p(stream.cuda_stream)))
                e1.record(stream)
                torch.cuda.synchronize()
                ms0.append(e0.elapsed_time(e1))
            L_.lib().qecmc_plan_destroy(plan0)
            m0 = float(np.mean(ms0[1:]))
            out["random_scan"] = {"note": "scan = 0 on the same batch: one generator pick per ladder and proposal, states in LDS (the kernel of rounds 1-3)",
                                  "kernel_ms_per_launch": m0, "proposals_per_s": proposals_per_pass / (m0 * 1e-3),
                                  "chain_sweeps_per_s": proposals_per_pass / n_gen / (m0 * 1e-3),
                                  "roofline_frac": algo_bytes / (m0 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and toric and args.scan == "random" and args.eta is None and args.sweep:
            # the library's second scan mode on the same batch, for the record (`value` above is the reference's chain)
            pr2 = L_.make_params(code=code_id, L=L, Nc=Nc, p=args.p, p_logical=args.p_logical, iters=args.iters,
                                 steps=args.ladder_steps, tops_burn=2, seed=args.seed, device=local_rank,
                                 scan=L_.SCAN_SWEEP)
            plan2 = C.c_void_p()
            L_.check(L_.lib().qecmc_plan_create(pr2, C.byref(plan2)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(2):
                e0.record(stream)
                L_.check(L_.lib().qecmc_pteq_launch_dev(plan2, sh.d_init.data_ptr(), N, first, d_counts.data_ptr(),
                                                        d_samples.data_ptr(), d_tops0.data_ptr(), None, None, None, None, 0,
                                                        C.c_void_p(stream.cuda_stream)))
                e1.record(stream)
                torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1)
            out["sweep_scan"] = {"note": "scan=1: systematic generator sweep (not the reference's chain; same stationary law, "
                                         "validated against exact enumeration), same batch and ladder steps",
                                 "kernel_ms_per_launch": ms2, "proposals_per_s": proposals_per_pass / (ms2 * 1e-3),
                                 "chain_sweeps_per_s": proposals_per_pass / n_gen / (ms2 * 1e-3)}
            L_.lib().qecmc_plan_destroy(plan2)
        print(json.dumps(out))
    sh.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
