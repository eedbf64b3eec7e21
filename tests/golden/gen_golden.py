#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the reference.

Runs ONLY in the build container (needs /root/reference; no-op elsewhere).  The
reference is pure Python + numba; numba is not installed here, so an
identity-decorator `numba` stub (created in a temp dir, outside the read-only
reference tree) lets the reference modules import and run as plain NumPy/Python
(SURVEY.md §8c).  Nothing from the reference is copied: the fixtures hold only
inputs and the reference's outputs.

    python tests/golden/gen_golden.py [--only f1,f2,f3,f4]

Fixtures
  f1_toric.npz    deterministic known-answer vectors for the toric stencils
  f2_toric.npz    stream-injected exact trajectories (Chain / Ladder / PTEQ);
                  uniform streams are random.Random(seed).random() so only the
                  seed is stored
  f3_toric.npz    replica-averaged PTEQ class histograms (statistical)
  f4_config1.npz  BASELINE config-1 plumbing vector
  f5_stats.npz    per-rung <n_errors>, per-pair swap acceptance and class counts after a 20 % burn-in: toric L=9 p=0.15 Nc=8,
                  rotated L=5/7 p=0.17, Ladder_biased xzzx L=5/7 eta=100 (statistical)
  f1_surf.npz / f2_surf.npz   the same for the XZZX and rotated codes, incl. the biased chain
  f2_alpha.npz                Chain_alpha / Ladder_alpha / PTEQ_alpha trajectories (src/mcmc_alpha.py)
  f_ptdc.npz                  PTDC_droplet unique-chain length histograms N(n) and PTDC percent vectors (decoders.py:138-233)
  f_convmult.npz              the conv_mult early stop of PTDC_droplet / STDC_droplet / STDC / STRC (decoders.py:153-162,:256-262,:783-826)
  f_xyz.npz                   STDC_droplet_general_noise / STDC_general_noise(_shortest) on the planar code (decoders.py:325-507), with
                              Chain and Chain_xyz (mcmc.py:106-114,162-173) sampling
  f_nalpha.npz                STDC_droplet_alpha / STDC_Nall_n_alpha (decoders.py:510-581) on the xzzx and rotated codes
  f_api_surface.npz           what decoders.py / decoders_biasednoise.py touch on the sampler path's classes: names, call shapes, result kinds (--only api)
  f_planar.npz                Planar_code stencil KATs and Chain (incl. update_chain_fast) / Ladder / PTEQ trajectories
"""
import argparse
import os
import random
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    shim = tempfile.mkdtemp(prefix="numba_stub_")
    os.makedirs(os.path.join(shim, "numba"))
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as f:
        f.write("def _deco(*a, **k):\n"
                "    if len(a) == 1 and callable(a[0]) and not k:\n"
                "        return a[0]\n"
                "    return lambda f: f\n"
                "njit = jit = _deco\n")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, shim)
    os.chdir(tempfile.mkdtemp(prefix="refcwd_"))
    import src.toric_model as tm
    import src.mcmc as mc
    import decoders as dec
    return tm, mc, dec


def import_reference_surf():
    import src.xzzx_model as xm
    import src.rotated_surface_model as rm
    import src.mcmc_biased as mb
    import decoders_biasednoise as decb
    return xm, rm, mb, decb


class Stream:
    """random.Random(seed).random with a draw counter; installed over the
    reference's two aliases of `random`."""

    def __init__(self, seed):
        self.r = random.Random(seed)
        self.n = 0

    def __call__(self):
        self.n += 1
        return self.r.random()


def install(stream, *models):
    random.random = stream          # `rand.random()` in src/mcmc*.py and the models
    for m in models:
        m.random = stream           # `from random import random` in the model modules


_ORIG_RANDOM = random.random


def restore(*models):
    random.random = _ORIG_RANDOM
    for m in models:
        m.random = _ORIG_RANDOM


def rand_matrix(rng, L, p):
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random((2, L, L)) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


# --------------------------------------------------------------------------- F1
def gen_f1(tm):
    rng = np.random.default_rng(20200915)
    out = {}
    for L in (3, 5, 9):
        for rep, p in enumerate((0.15, 0.45)):
            tag = f"L{L}_{rep}"
            m = rand_matrix(rng, L, p)
            code = tm.Toric_code(L)
            code.qubit_matrix = m.copy()
            out[f"{tag}_m"] = m
            out[f"{tag}_count"] = np.int64(code.count_errors())
            out[f"{tag}_class"] = np.int64(code.define_equivalence_class())
            code.syndrom()
            out[f"{tag}_defects"] = np.asarray(code.defect_matrix, dtype=np.uint8)
            st_new, st_dE = [], []
            for op in (1, 3):
                for r in range(L):
                    for c in range(L):
                        new, dE = code.apply_stabilizer(r, c, op)
                        st_new.append(new); st_dE.append(dE)
            out[f"{tag}_stab_new"] = np.array(st_new, dtype=np.uint8)     # [2*L*L,2,L,L] order (op,r,c)
            out[f"{tag}_stab_dE"] = np.array(st_dE, dtype=np.int64)
            lg_new, lg_dE, lg_arg = [], [], []
            for op in range(4):
                for layer in (0, 1):
                    for xp in range(L):
                        for zp in range(L):
                            new, dE = tm._apply_logical(m, op, layer, xp, zp)
                            lg_new.append(new); lg_dE.append(dE); lg_arg.append((op, layer, xp, zp))
            out[f"{tag}_log_new"] = np.array(lg_new, dtype=np.uint8)
            out[f"{tag}_log_dE"] = np.array(lg_dE, dtype=np.int64)
            out[f"{tag}_log_arg"] = np.array(lg_arg, dtype=np.int64)
            out[f"{tag}_to_class"] = np.array([code.to_class(eq) for eq in range(16)], dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "f1_toric.npz"), **out)
    print("f1_toric.npz", len(out), "arrays")


# --------------------------------------------------------------------------- F2
def gen_f2(tm, mc, dec):
    rng = np.random.default_rng(7)
    out = {}
    cases = []
    # Chain.update_chain: non-top and top variants (src/mcmc.py:19-43)
    for i, (L, p, p_logical, iters, perr) in enumerate([
            (3, 0.5, 0.0, 25, 0.3), (5, 0.10, 0.0, 400, 0.1), (9, 0.15, 0.0, 600, 0.15),
            (9, 0.40, 0.0, 600, 0.15), (3, 0.75, 0.5, 25, 0.3), (5, 0.75, 0.5, 300, 0.1),
            (9, 0.75, 0.5, 300, 0.15), (5, 0.30, 0.5, 300, 0.1), (9, 0.20, 0.25, 300, 0.15)]):
        m = rand_matrix(rng, L, perr)
        seed = 1000 + i
        code = tm.Toric_code(L); code.qubit_matrix = m.copy()
        ch = mc.Chain(p, code); ch.p_logical = p_logical
        s = Stream(seed); install(s, tm)
        ch.update_chain(iters)
        restore(tm)
        tag = f"chain{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([L, p, p_logical, iters, seed, s.n], dtype=np.float64)
        cases.append(tag)
    # Ladder.step (src/mcmc.py:94-103)
    for i, (L, p, Nc, iters, nstep, perr) in enumerate([
            (3, 0.3, 4, 5, 40, 0.3), (5, 0.10, 5, 10, 60, 0.10), (9, 0.15, 8, 10, 40, 0.15),
            (5, 0.25, 3, 7, 50, 0.2), (3, 0.05, 2, 10, 60, 0.2)]):
        m = rand_matrix(rng, L, perr)
        seed = 2000 + i
        code = tm.Toric_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm)
        ld = mc.Ladder(p, code, Nc, 0.5)
        tops_hist = []
        for _ in range(nstep):
            ld.step(iters)
            tops_hist.append(ld.tops0)
        restore(tm)
        tag = f"ladder{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_states"] = np.array([c.code.qubit_matrix for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_flags"] = np.array([c.flag for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_tops_hist"] = np.array(tops_hist, dtype=np.int64)
        out[f"{tag}_p_ladder"] = np.asarray(ld.p_ladder, dtype=np.float64)
        out[f"{tag}_p_diff"] = np.asarray(ld.p_diff, dtype=np.float64)
        out[f"{tag}_par"] = np.array([L, p, Nc, iters, nstep, seed, s.n], dtype=np.float64)
        cases.append(tag)
    # decoders.PTEQ (decoders.py:25-89), fixed steps and the error_based criterion
    for i, (L, p, Nc, iters, steps, tops_burn, conv, perr, SEQ, TOPS, eps) in enumerate([
            (3, 0.10, 3, 10, 400, 2, None, 0.15, 2, 10, 0.1), (3, 0.10, 3, 10, 400, 0, None, 0.15, 2, 10, 0.1),
            (5, 0.10, 5, 10, 300, 1, None, 0.10, 2, 10, 0.1), (9, 0.15, 8, 10, 60, 0, None, 0.15, 2, 10, 0.1),
            (3, 0.10, 3, 10, 4000, 2, "error_based", 0.15, 2, 10, 0.1),
            (3, 0.05, 3, 5, 6000, 2, "error_based", 0.1, 2, 10, 0.3),
            (3, 0.10, 3, 10, 6000, 1, "error_based", 0.1, 1, 4, 0.5),
            (5, 0.05, 4, 10, 6000, 2, "error_based", 0.05, 2, 6, 0.4),
            (5, 0.10, 5, 10, 8000, 2, "error_based", 0.10, 2, 10, 0.25)]):
        m = rand_matrix(rng, L, perr)
        seed = 3000 + i
        code = tm.Toric_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm)
        pct = dec.PTEQ(code, p, Nc=Nc, SEQ=SEQ, TOPS=TOPS, eps=eps, steps=steps, iters=iters,
                       tops_burn=tops_burn, conv_criteria=conv)
        restore(tm)
        tag = f"pteq{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_percent"] = np.asarray(pct, dtype=np.uint8)
        out[f"{tag}_par"] = np.array([L, p, Nc, iters, steps, tops_burn, 1 if conv else 0, seed, s.n,
                                      SEQ, TOPS, eps], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f2_toric.npz"), **out)
    print("f2_toric.npz", cases)


# --------------------------------------------------------------------------- F1/F2 for xzzx + rotated
def rand_matrix2(rng, L, p):
    m = np.zeros((L, L), dtype=np.uint8)
    err = rng.random((L, L)) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


def _surf_defects(mod, m):
    """the loops of xzzx_code.syndrome / RotSurCode.syndrome (without the plot call)"""
    L = m.shape[0]
    d = np.zeros((L + 1, L + 1), dtype=np.uint8)
    for i in range(L - 1):
        for j in range(L - 1):
            d[i + 1, j + 1] = mod._find_syndrome(m, i, j, 1)
    for i in range(int((L - 1) / 2)):
        for j, (r, c) in enumerate([(0, 2 * i + 2), (2 * i + 2, L), (L, 2 * i + 1), (2 * i + 1, 0)]):
            d[r, c] = mod._find_syndrome(m, i, j, 3)
    return d


def gen_f1_surf(xm, rm):
    rng = np.random.default_rng(424242)
    out = {}
    for name, mod, cls in (("xzzx", xm, xm.xzzx_code), ("rot", rm, rm.RotSurCode)):
        for L in (3, 5, 9):
            for rep, p in enumerate((0.15, 0.45)):
                t = f"{name}_L{L}_{rep}"
                m = rand_matrix2(rng, L, p)
                code = cls(L); code.qubit_matrix = m.copy()
                out[f"{t}_m"] = m
                out[f"{t}_count"] = np.int64(code.count_errors())
                out[f"{t}_class"] = np.int64(code.define_equivalence_class())
                out[f"{t}_defects"] = _surf_defects(mod, m)
                arg, new, dE = [], [], []
                for r in range(L - 1):
                    for c in range(L - 1):
                        n, d = code.apply_stabilizer(r, c, 1); arg.append((r, c, 1)); new.append(n); dE.append(d)
                for r in range((L - 1) // 2):
                    for c in range(4):
                        n, d = code.apply_stabilizer(r, c, 3); arg.append((r, c, 3)); new.append(n); dE.append(d)
                out[f"{t}_stab_arg"] = np.array(arg, dtype=np.int64)
                out[f"{t}_stab_new"] = np.array(new, dtype=np.uint8)
                out[f"{t}_stab_dE"] = np.array(dE, dtype=np.int64)
                arg, new, dE, cl = [], [], [], []
                for op in range(4):
                    for xp in range(L):
                        for zp in range(L):
                            n, d = code.apply_logical(op, xp, zp)
                            arg.append((op, xp, zp)); new.append(n); dE.append(d)
                            c2 = cls(L); c2.qubit_matrix = n; cl.append(c2.define_equivalence_class())
                out[f"{t}_log_arg"] = np.array(arg, dtype=np.int64)
                out[f"{t}_log_new"] = np.array(new, dtype=np.uint8)
                out[f"{t}_log_dE"] = np.array(dE, dtype=np.int64)
                out[f"{t}_log_class"] = np.array(cl, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "f1_surf.npz"), **out)
    print("f1_surf.npz", len(out), "arrays")


def gen_f2_surf(xm, rm, mc, mb, dec, decb):
    rng = np.random.default_rng(99)
    out = {}
    cases = []
    mods = (xm, rm)
    codes = {"xzzx": xm.xzzx_code, "rot": rm.RotSurCode}
    # depolarizing Chain (src/mcmc.py) on both codes
    for i, (name, L, p, p_logical, iters, perr) in enumerate([
            ("rot", 3, 0.5, 0.0, 40, 0.3), ("rot", 5, 0.17, 0.0, 400, 0.15), ("rot", 9, 0.17, 0.0, 500, 0.15),
            ("rot", 5, 0.75, 0.5, 300, 0.15), ("rot", 9, 0.3, 0.5, 300, 0.15), ("xzzx", 5, 0.15, 0.0, 400, 0.15),
            ("xzzx", 9, 0.75, 0.5, 300, 0.15), ("xzzx", 7, 0.2, 0.5, 200, 0.2), ("rot", 7, 0.1, 0.0, 200, 0.2)]):
        m = rand_matrix2(rng, L, perr)
        seed = 5000 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        ch = mc.Chain(p, code); ch.p_logical = p_logical
        s = Stream(seed); install(s, *mods)
        ch.update_chain(iters)
        restore(*mods)
        tag = f"chain{i}"
        out[f"{tag}_init"] = m; out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, p, p_logical, iters, seed, s.n, 0, 0], dtype=np.float64)
        cases.append(tag)
    # Chain_biased (src/mcmc_biased.py)
    for i, (name, L, p, eta, p_logical, iters, perr) in enumerate([
            ("xzzx", 3, 0.3, 10, 0.0, 60, 0.3), ("xzzx", 5, 0.15, 100, 0.0, 300, 0.15), ("xzzx", 9, 0.15, 100, 0.0, 300, 0.15),
            ("xzzx", 5, 0.4, 10, 0.5, 300, 0.15), ("xzzx", 9, 0.5, 100, 0.5, 200, 0.15), ("rot", 5, 0.2, 3, 0.5, 200, 0.15)]):
        m = rand_matrix2(rng, L, perr)
        seed = 5100 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        ch = mb.Chain_biased(p, eta, code); ch.p_logical = p_logical
        s = Stream(seed); install(s, *mods)
        ch.update_chain(iters)
        restore(*mods)
        tag = f"bchain{i}"
        out[f"{tag}_init"] = m; out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, p, p_logical, iters, seed, s.n, 1, eta], dtype=np.float64)
        cases.append(tag)
    # Ladder / Ladder_biased
    for i, (name, biased, L, p, eta, Nc, iters, nstep, perr) in enumerate([
            ("rot", 0, 3, 0.3, 0, 4, 5, 40, 0.3), ("rot", 0, 5, 0.17, 0, 5, 10, 50, 0.15), ("rot", 0, 9, 0.17, 0, 8, 10, 30, 0.15),
            ("xzzx", 0, 5, 0.15, 0, 4, 10, 40, 0.15), ("xzzx", 1, 3, 0.3, 10, 3, 5, 40, 0.3), ("xzzx", 1, 5, 0.15, 100, 5, 10, 40, 0.15),
            ("xzzx", 1, 9, 0.15, 100, 8, 10, 20, 0.15)]):
        m = rand_matrix2(rng, L, perr)
        seed = 5200 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, *mods)
        ld = mb.Ladder_biased(p, code, eta, Nc, 0.5) if biased else mc.Ladder(p, code, Nc, 0.5)
        tops = []
        for _ in range(nstep):
            ld.step(iters); tops.append(ld.tops0)
        restore(*mods)
        tag = f"ladder{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_states"] = np.array([c.code.qubit_matrix for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_flags"] = np.array([c.flag for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_tops_hist"] = np.array(tops, dtype=np.int64)
        out[f"{tag}_p_ladder"] = np.asarray(ld.p_ladder, dtype=np.float64)
        out[f"{tag}_p_diff"] = np.asarray(ld.p_diff, dtype=np.float64)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, p, Nc, iters, nstep, seed, s.n, biased, eta], dtype=np.float64)
        cases.append(tag)
    # PTEQ (rotated, depolarizing) and PTEQ_biased (xzzx)
    for i, (name, biased, L, p, eta, Nc, iters, steps, tops_burn, conv, perr, SEQ, TOPS, eps) in enumerate([
            ("rot", 0, 3, 0.17, 0, 3, 10, 300, 2, None, 0.15, 2, 10, 0.1), ("rot", 0, 5, 0.17, 0, 5, 10, 200, 0, None, 0.15, 2, 10, 0.1),
            ("rot", 0, 3, 0.17, 0, 3, 10, 6000, 1, "error_based", 0.15, 1, 4, 0.5),
            ("xzzx", 1, 3, 0.3, 10, 3, 10, 300, 2, None, 0.3, 2, 10, 0.1), ("xzzx", 1, 5, 0.15, 100, 5, 10, 150, 0, None, 0.15, 2, 10, 0.1),
            ("xzzx", 1, 5, 0.15, 100, 5, 10, 4000, 2, "error_based", 0.15, 2, 10, 0.1)]):
        m = rand_matrix2(rng, L, perr)
        seed = 5300 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, *mods)
        if biased:
            pct = decb.PTEQ_biased(code, p, eta=eta, Nc=Nc, SEQ=SEQ, TOPS=TOPS, eps=eps, steps=steps, iters=iters,
                                   tops_burn=tops_burn, conv_criteria=conv)
        else:
            pct = dec.PTEQ(code, p, Nc=Nc, SEQ=SEQ, TOPS=TOPS, eps=eps, steps=steps, iters=iters, tops_burn=tops_burn,
                           conv_criteria=conv)
        restore(*mods)
        tag = f"pteq{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_percent"] = np.asarray(pct, dtype=np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, p, Nc, iters, steps, tops_burn, 1 if conv else 0, seed, s.n,
                                      SEQ, TOPS, eps, biased, eta], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f2_surf.npz"), **out)
    print("f2_surf.npz", cases)


# --------------------------------------------------------------------------- F3
def gen_f2_alpha(xm, rm, ma, decb):
    """Chain_alpha / Ladder_alpha (src/mcmc_alpha.py) and PTEQ_alpha (decoders_biasednoise.py:175) on the injected stream."""
    rng = np.random.default_rng(177)
    out = {}
    cases = []
    mods = (xm, rm)
    codes = {"xzzx": xm.xzzx_code, "rot": rm.RotSurCode}
    for i, (name, L, pzt, alpha, p_logical, iters, perr) in enumerate([
            ("xzzx", 3, 0.2, 2.0, 0.0, 60, 0.3), ("xzzx", 5, 0.1, 1.7, 0.0, 300, 0.15), ("xzzx", 9, 0.08, 2.5, 0.0, 300, 0.15),
            ("xzzx", 5, 1.0, 2.0, 0.5, 300, 0.15), ("xzzx", 9, 0.5, 1.3, 0.5, 200, 0.15), ("rot", 5, 0.15, 1.0, 0.5, 200, 0.15),
            ("rot", 7, 0.12, 3.1, 0.0, 300, 0.15)]):
        m = rand_matrix2(rng, L, perr)
        seed = 6100 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        ch = ma.Chain_alpha(np.float64(pzt), alpha, code); ch.p_logical = p_logical
        s = Stream(seed); install(s, *mods)
        ch.update_chain(iters)
        restore(*mods)
        tag = f"achain{i}"
        out[f"{tag}_init"] = m; out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, pzt, p_logical, iters, seed, s.n, alpha, ch.n_eff], dtype=np.float64)
        cases.append(tag)
    for i, (name, L, pzt, alpha, Nc, iters, nstep, perr) in enumerate([
            ("xzzx", 3, 0.2, 2.0, 3, 5, 60, 0.3), ("xzzx", 5, 0.1, 1.7, 5, 10, 60, 0.15), ("xzzx", 9, 0.08, 2.5, 8, 10, 30, 0.15),
            ("rot", 5, 0.15, 1.0, 4, 10, 60, 0.15), ("rot", 7, 0.12, 3.1, 6, 10, 40, 0.15), ("xzzx", 7, 0.05, 1.4, 7, 10, 40, 0.1)]):
        m = rand_matrix2(rng, L, perr)
        seed = 6200 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, *mods)
        ld = ma.Ladder_alpha(pzt, code, alpha, Nc, 0.5)
        tops = []; neff = []
        for _ in range(nstep):
            ld.step(iters); tops.append(ld.tops0); neff.append([c.n_eff for c in ld.chains])
        restore(*mods)
        tag = f"aladder{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_states"] = np.array([c.code.qubit_matrix for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_flags"] = np.array([c.flag for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_tops_hist"] = np.array(tops, dtype=np.int64)
        out[f"{tag}_neff_hist"] = np.array(neff, dtype=np.float64)
        out[f"{tag}_p_ladder"] = np.asarray(ld.pz_tilde_ladder, dtype=np.float64)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, pzt, Nc, iters, nstep, seed, s.n, alpha], dtype=np.float64)
        cases.append(tag)
    for i, (name, L, pzt, alpha, Nc, iters, steps, tops_burn, conv, perr, SEQ, TOPS, eps) in enumerate([
            ("xzzx", 3, 0.2, 2.0, 3, 10, 300, 2, None, 0.3, 2, 10, 0.1), ("xzzx", 5, 0.1, 1.7, 5, 10, 200, 0, None, 0.15, 2, 10, 0.1),
            ("rot", 5, 0.15, 1.3, 5, 10, 200, 1, None, 0.15, 2, 10, 0.1),
            ("xzzx", 3, 0.2, 2.0, 3, 10, 6000, 1, "error_based", 0.3, 1, 4, 0.5),
            ("xzzx", 5, 0.1, 1.7, 5, 10, 4000, 2, "error_based", 0.15, 2, 10, 0.1)]):
        m = rand_matrix2(rng, L, perr)
        seed = 6300 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, *mods)
        pct = decb.PTEQ_alpha(code, pzt, alpha=alpha, Nc=Nc, SEQ=SEQ, TOPS=TOPS, eps=eps, steps=steps, iters=iters,
                              tops_burn=tops_burn, conv_criteria=conv)
        restore(*mods)
        tag = f"apteq{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_percent"] = np.asarray(pct, dtype=np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, pzt, Nc, iters, steps, tops_burn, 1 if conv else 0, seed, s.n,
                                      SEQ, TOPS, eps, alpha], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f2_alpha.npz"), **out)
    print("f2_alpha.npz", cases)


def rand_planar(rng, L, p):
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random((2, L, L)) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    m[1, -1, :] = 0                      # planar_model.py:38-39
    m[1, :, -1] = 0
    return m


def gen_planar(pm, mc, dec):
    """Planar_code (src/planar_model.py): every stabilizer / logical, class, syndrome (exact KATs) and Chain
    (update_chain and update_chain_fast, which is hard-wired to this stencil, mcmc.py:6,152-160), Ladder and PTEQ
    trajectories on the injected stream."""
    rng = np.random.default_rng(404)
    out = {}
    kats = []
    for L in (3, 4, 7):
        for rep in range(2):
            m = rand_planar(rng, L, 0.25)
            code = pm.Planar_code(L); code.qubit_matrix = m.copy()
            t = f"L{L}_{rep}"
            out[f"{t}_m"] = m
            out[f"{t}_count"] = np.int64(code.count_errors())
            out[f"{t}_class"] = np.int64(code.define_equivalence_class())
            code.syndrom()
            out[f"{t}_vertex"] = np.asarray(code.vertex_defects, dtype=np.uint8)
            out[f"{t}_plaquette"] = np.asarray(code.plaquette_defects, dtype=np.uint8)
            args, news, dEs = [], [], []
            for r in range(L - 1):
                for c in range(L):
                    for (rr, cc, op) in ((r, c, 1), (c, r, 3)):     # op 1: (short, long); op 3: (long, short), planar_model.py:347-352
                        new, dE = code.apply_stabilizer(rr, cc, op)
                        args.append((rr, cc, op)); news.append(new.copy()); dEs.append(dE)
            out[f"{t}_stab_arg"] = np.array(args, dtype=np.int64); out[f"{t}_stab_new"] = np.array(news, dtype=np.uint8)
            out[f"{t}_stab_dE"] = np.array(dEs, dtype=np.int64)
            args, news, dEs, cls = [], [], [], []
            for op in range(4):
                for xp in range(L):
                    for zp in range(L):
                        new, dE = code.apply_logical(op, xp, zp)
                        args.append((op, xp, zp)); news.append(new.copy()); dEs.append(dE)
                        c2 = pm.Planar_code(L); c2.qubit_matrix = new
                        cls.append(c2.define_equivalence_class())
            out[f"{t}_log_arg"] = np.array(args, dtype=np.int64); out[f"{t}_log_new"] = np.array(news, dtype=np.uint8)
            out[f"{t}_log_dE"] = np.array(dEs, dtype=np.int64); out[f"{t}_log_class"] = np.array(cls, dtype=np.int64)
            kats.append(t)
    out["kats"] = np.array(kats)
    cases = []
    for i, (L, p, p_logical, iters, perr, fast) in enumerate([
            (3, 0.5, 0.0, 60, 0.3, 0), (5, 0.15, 0.0, 400, 0.15, 0), (7, 0.12, 0.0, 400, 0.12, 1), (5, 0.75, 0.5, 300, 0.15, 0),
            (7, 0.3, 0.5, 300, 0.15, 0), (4, 0.2, 0.0, 200, 0.2, 1)]):
        m = rand_planar(rng, L, perr)
        seed = 7000 + i
        code = pm.Planar_code(L); code.qubit_matrix = m.copy()
        ch = mc.Chain(p, code); ch.p_logical = p_logical
        s = Stream(seed); install(s, pm, mc)
        if fast:
            for _ in range(iters // 5):
                ch.update_chain_fast(5)                 # STDC's inner call, decoders.py:250
        else:
            ch.update_chain(iters)
        restore(pm, mc)
        tag = f"chain{i}"
        out[f"{tag}_init"] = m; out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([L, p, p_logical, iters, seed, s.n, fast], dtype=np.float64)
        cases.append(tag)
    for i, (L, p, Nc, iters, nstep, perr) in enumerate([(3, 0.3, 3, 5, 50, 0.3), (5, 0.15, 5, 10, 50, 0.15), (7, 0.12, 8, 10, 25, 0.12)]):
        m = rand_planar(rng, L, perr)
        seed = 7100 + i
        code = pm.Planar_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, pm, mc)
        ld = mc.Ladder(p, code, Nc, 0.5)
        tops = []
        for _ in range(nstep):
            ld.step(iters); tops.append(ld.tops0)
        restore(pm, mc)
        tag = f"ladder{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_states"] = np.array([c.code.qubit_matrix for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_flags"] = np.array([c.flag for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_tops_hist"] = np.array(tops, dtype=np.int64)
        out[f"{tag}_par"] = np.array([L, p, Nc, iters, nstep, seed, s.n], dtype=np.float64)
        cases.append(tag)
    for i, (L, p, Nc, iters, steps, tops_burn, conv, perr, SEQ, TOPS, eps) in enumerate([
            (3, 0.17, 3, 10, 300, 2, None, 0.15, 2, 10, 0.1), (5, 0.12, 5, 10, 200, 0, None, 0.12, 2, 10, 0.1),
            (3, 0.17, 3, 10, 6000, 1, "error_based", 0.15, 1, 4, 0.5)]):
        m = rand_planar(rng, L, perr)
        seed = 7200 + i
        code = pm.Planar_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, pm, mc)
        pct = dec.PTEQ(code, p, Nc=Nc, SEQ=SEQ, TOPS=TOPS, eps=eps, steps=steps, iters=iters, tops_burn=tops_burn, conv_criteria=conv)
        restore(pm, mc)
        tag = f"pteq{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_percent"] = np.asarray(pct, dtype=np.uint8)
        out[f"{tag}_par"] = np.array([L, p, Nc, iters, steps, tops_burn, 1 if conv else 0, seed, s.n, SEQ, TOPS, eps], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f_planar.npz"), **out)
    print("f_planar.npz", kats, cases)


def gen_ptdc(tm, pm, mc, dec):
    """PTDC_droplet (decoders.py:138-164): the unique-chain set of a class ladder -> N(n), the number of distinct chains of
    each length; and PTDC's percent vector (droplets = 1) for a toric syndrome."""
    rng = np.random.default_rng(808)
    out = {}
    cases = []
    for i, (name, L, p, Nc, steps, perr) in enumerate([("toric", 3, 0.1, 3, 300, 0.15), ("toric", 5, 0.1, 5, 200, 0.1),
                                                      ("planar", 3, 0.15, 3, 300, 0.15), ("planar", 5, 0.1, 4, 150, 0.1),
                                                      ("toric", 4, 0.3, 2, 200, 0.2)]):
        m = rand_matrix(rng, L, perr) if name == "toric" else rand_planar(rng, L, perr)
        seed = 8000 + i
        code = tm.Toric_code(L) if name == "toric" else pm.Planar_code(L)
        code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm, pm, mc)
        ld = mc.Ladder(p, code, Nc)                       # as PTDC builds it (decoders.py:182): no p_logical
        samples = dec.PTDC_droplet(ld, steps, 10, 0)
        restore(tm, pm, mc)
        nq = m.size
        tag = f"drop{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_hist"] = np.bincount(np.array(list(samples.values()), dtype=np.int64), minlength=nq + 1).astype(np.uint32)
        out[f"{tag}_states"] = np.array([c.code.qubit_matrix for c in ld.chains], dtype=np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "toric" else 3, L, p, Nc, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    for i, (L, p_error, p_sampling, Nc, steps, perr) in enumerate([(3, 0.1, None, 3, 600, 0.15), (3, 0.08, 0.2, 2, 400, 0.1)]):
        m = rand_matrix(rng, L, perr)
        seed = 8100 + i
        code = tm.Toric_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm, pm, mc)
        pct = dec.PTDC(code, p_error, p_sampling=p_sampling, droplets=1, Nc=Nc, steps=steps, conv_mult=0)
        restore(tm, pm, mc)
        tag = f"ptdc{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_percent"] = np.asarray(pct, dtype=np.uint8)
        out[f"{tag}_classes"] = np.array([tm._to_class(eq, m.copy()) for eq in range(16)], dtype=np.uint8)   # toric_model.py:354
        out[f"{tag}_par"] = np.array([L, p_error, p_sampling or p_error, Nc, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    # STDC_droplet / STDC (decoders.py:236-322) on the planar code: single chains, update_chain_fast(5) per step
    for i, (L, p, steps, perr) in enumerate([(3, 0.15, 400, 0.15), (5, 0.1, 300, 0.1)]):
        m = rand_planar(rng, L, perr)
        seed = 8200 + i
        code = pm.Planar_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm, pm, mc)
        ch = mc.Chain(p, code)
        samples = dec.STDC_droplet(ch, steps, False, 0)
        restore(tm, pm, mc)
        tag = f"sdrop{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_hist"] = np.bincount(np.array(list(samples.values()), dtype=np.int64), minlength=m.size + 1).astype(np.uint32)
        out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([L, p, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    for i, (L, p_error, p_sampling, steps, perr) in enumerate([(3, 0.1, None, 500, 0.15), (4, 0.08, 0.2, 300, 0.1)]):
        m = rand_planar(rng, L, perr)
        seed = 8300 + i
        inits = []
        for op in range(4):                                   # one representative per class (the list form, :273-280)
            c = pm.Planar_code(L); c.qubit_matrix, _ = pm._apply_logical(m.copy(), op, 0, 0)
            inits.append(c)
        inits.sort(key=lambda c: c.define_equivalence_class())
        s = Stream(seed); install(s, tm, pm, mc)
        dist = dec.STDC(inits, p_error, p_sampling=p_sampling, droplets=1, steps=steps, conv_mult=0)
        restore(tm, pm, mc)
        tag = f"stdc{i}"
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_dist"] = np.asarray(dist, dtype=np.float64)
        out[f"{tag}_par"] = np.array([L, p_error, p_sampling or p_error, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    # STRC (decoders.py:745-949, planar list form) and PTRC (:584-742, toric list form): curve-fit estimators on N(n), m(n)
    for i, (L, p_error, p_sampling, steps, perr) in enumerate([(3, 0.1, None, 500, 0.15), (4, 0.08, 0.25, 400, 0.1)]):
        m = rand_planar(rng, L, perr)
        seed = 8400 + i
        inits = []
        for op in range(4):
            c = pm.Planar_code(L); c.qubit_matrix, _ = pm._apply_logical(m.copy(), op, 0, 0)
            inits.append(c)
        inits.sort(key=lambda c: c.define_equivalence_class())
        s = Stream(seed); install(s, tm, pm, mc)
        dist = dec.STRC(inits, p_error, p_sampling=p_sampling, droplets=1, steps=steps, conv_mult=0)
        restore(tm, pm, mc)
        tag = f"strc{i}"
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_dist"] = np.asarray(dist, dtype=np.float64)
        out[f"{tag}_par"] = np.array([L, p_error, p_sampling or p_error, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    for i, (L, p_error, p_sampling, Nc, steps, perr) in enumerate([(3, 0.1, None, 3, 900, 0.15), (3, 0.08, 0.2, 4, 800, 0.1)]):
        m = rand_matrix(rng, L, perr)
        seed = 8500 + i
        inits = []
        for eq in range(16):
            c = tm.Toric_code(L); c.qubit_matrix = tm._to_class(eq, m.copy()); inits.append(c)
        s = Stream(seed); install(s, tm, pm, mc)
        pct = dec.PTRC(inits, p_error, p_sampling=p_sampling, droplets=1, Nc=Nc, steps=steps)
        restore(tm, pm, mc)
        tag = f"ptrc{i}"
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_percent"] = np.asarray(pct, dtype=np.uint8)
        out[f"{tag}_par"] = np.array([L, p_error, p_sampling or p_error, Nc, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    # single_temp (decoders.py:108-135): mean chain length per class from single chains, update_chain_fast(5) per step
    for i, (L, p, max_iters, perr) in enumerate([(3, 0.12, 300, 0.15), (5, 0.2, 200, 0.1)]):
        m = rand_planar(rng, L, perr)
        seed = 8600 + i
        inits = []
        for op in range(4):
            c = pm.Planar_code(L); c.qubit_matrix, _ = pm._apply_logical(m.copy(), op, 0, 0)
            inits.append(c)
        inits.sort(key=lambda c: c.define_equivalence_class())
        s = Stream(seed); install(s, tm, pm, mc)
        means = dec.single_temp(inits, p, max_iters)
        restore(tm, pm, mc)
        tag = f"stemp{i}"
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_means"] = np.asarray(means, dtype=np.float64)
        out[f"{tag}_par"] = np.array([L, p, max_iters, seed, s.n], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f_ptdc.npz"), **out)
    print("f_ptdc.npz", cases)


def gen_convmult(tm, pm, mc, dec):
    """The conv_mult early stop (decoders.py:153-162, :256-262, :783-826): same set-ups as f_ptdc with conv_mult != 0; the number
    of draws consumed pins the step at which each droplet stopped."""
    rng = np.random.default_rng(909)
    out = {}
    cases = []
    for i, (name, L, p, Nc, steps, perr, cm) in enumerate([("toric", 3, 0.1, 3, 400, 0.15, 2.0), ("toric", 5, 0.1, 5, 300, 0.1, 2.0),
                                                          ("planar", 3, 0.15, 3, 500, 0.15, 3.0), ("planar", 5, 0.1, 4, 300, 0.1, 1.5),
                                                          ("toric", 4, 0.3, 2, 1000, 0.2, 2.0)]):
        m = rand_matrix(rng, L, perr) if name == "toric" else rand_planar(rng, L, perr)
        seed = 9000 + i
        code = tm.Toric_code(L) if name == "toric" else pm.Planar_code(L)
        code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm, pm, mc)
        ld = mc.Ladder(p, code, Nc)
        samples = dec.PTDC_droplet(ld, steps, 10, cm)
        restore(tm, pm, mc)
        tag = f"drop{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_hist"] = np.bincount(np.array(list(samples.values()), dtype=np.int64), minlength=m.size + 1).astype(np.uint32)
        out[f"{tag}_par"] = np.array([0 if name == "toric" else 3, L, p, Nc, steps, seed, s.n, cm], dtype=np.float64)
        cases.append(tag)
    for i, (L, p, steps, perr, cm) in enumerate([(3, 0.15, 600, 0.15, 2.0), (5, 0.1, 800, 0.1, 2.5)]):
        m = rand_planar(rng, L, perr)
        seed = 9200 + i
        code = pm.Planar_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm, pm, mc)
        ch = mc.Chain(p, code)
        samples = dec.STDC_droplet(ch, steps, False, cm)
        restore(tm, pm, mc)
        tag = f"sdrop{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_hist"] = np.bincount(np.array(list(samples.values()), dtype=np.int64), minlength=m.size + 1).astype(np.uint32)
        out[f"{tag}_par"] = np.array([L, p, steps, seed, s.n, cm], dtype=np.float64)
        cases.append(tag)
    for i, (fn, L, p_error, p_sampling, steps, perr, cm) in enumerate([("STDC", 3, 0.1, None, 600, 0.15, 2.0),
                                                                      ("STRC", 3, 0.1, 0.2, 800, 0.15, 2.0),
                                                                      ("STRC", 4, 0.08, 0.25, 600, 0.1, 3.0)]):
        m = rand_planar(rng, L, perr)
        seed = 9300 + i
        inits = []
        for op in range(4):
            c = pm.Planar_code(L); c.qubit_matrix, _ = pm._apply_logical(m.copy(), op, 0, 0)
            inits.append(c)
        inits.sort(key=lambda c: c.define_equivalence_class())
        s = Stream(seed); install(s, tm, pm, mc)
        dist = getattr(dec, fn)(inits, p_error, p_sampling=p_sampling, droplets=1, steps=steps, conv_mult=cm)
        restore(tm, pm, mc)
        tag = f"{fn.lower()}{i}"
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_dist"] = np.asarray(dist, dtype=np.float64)
        out[f"{tag}_par"] = np.array([L, p_error, p_sampling or p_error, steps, seed, s.n, cm], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f_convmult.npz"), **out)
    print("f_convmult.npz", cases, {c: int(out[c + "_par"][-2]) for c in cases})


def gen_xyz(tm, pm, mc, dec):
    """STDC_droplet_general_noise (decoders.py:325-342): the (n_x, n_y, n_z) of every distinct chain a single planar chain
    visits, sampled by Chain (scalar p_sampling) or Chain_xyz (array p_sampling, mcmc.py:106-114,162-173); and the estimates
    STDC_general_noise / STDC_general_noise_shortest form from them (droplets = 1, list form of init_code)."""
    rng = np.random.default_rng(1010)
    out = {}
    cases = []
    for i, (L, p, steps, perr) in enumerate([(3, 0.15, 300, 0.15), (5, 0.1, 250, 0.1), (3, np.array([0.02, 0.05, 0.11]), 300, 0.15),
                                            (5, np.array([0.08, 0.01, 0.03]), 250, 0.1), (4, np.array([0.3, 0.2, 0.1]), 200, 0.2)]):
        m = rand_planar(rng, L, perr)
        seed = 10000 + i
        code = pm.Planar_code(L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, tm, pm, mc)
        ch = mc.Chain_xyz(p, code) if isinstance(p, np.ndarray) else mc.Chain(p, code)
        samples = dec.STDC_droplet_general_noise(ch, steps, False)
        restore(tm, pm, mc)
        tag = f"gdrop{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_xyz"] = np.array(list(samples.values()), dtype=np.int64).reshape(-1, 3)
        out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_p"] = np.atleast_1d(np.asarray(p, dtype=np.float64))
        out[f"{tag}_par"] = np.array([L, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    for i, (L, p_xyz, p_sampling, steps, perr) in enumerate([(3, np.array([0.03, 0.02, 0.08]), None, 400, 0.15),
                                                             (4, np.array([0.01, 0.01, 0.1]), 0.2, 300, 0.1),
                                                             (3, np.array([0.05, 0.03, 0.04]), np.array([0.1, 0.06, 0.08]), 400, 0.15)]):
        m = rand_planar(rng, L, perr)
        seed = 10100 + i
        inits = []
        for op in range(4):
            c = pm.Planar_code(L); c.qubit_matrix, _ = pm._apply_logical(m.copy(), op, 0, 0)
            inits.append(c)
        inits.sort(key=lambda c: c.define_equivalence_class())
        tag = f"gn{i}"
        for name, fn, kw in [("all", dec.STDC_general_noise, {}), ("short", dec.STDC_general_noise, dict(shortest_only=True)),
                             ("both", dec.STDC_general_noise_shortest, {})]:
            s = Stream(seed); install(s, tm, pm, mc)
            dist = fn(inits, p_xyz, p_sampling=p_sampling, droplets=1, steps=steps, **kw)
            restore(tm, pm, mc)
            out[f"{tag}_{name}"] = np.asarray(dist, dtype=np.float64)
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_pxyz"] = p_xyz
        out[f"{tag}_ps"] = np.atleast_1d(np.asarray(p_xyz.sum() if p_sampling is None else p_sampling, dtype=np.float64))
        out[f"{tag}_par"] = np.array([L, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f_xyz.npz"), **out)
    print("f_xyz.npz", cases, {c: int(out[c + "_par"][-1]) for c in cases})


def gen_nalpha(xm, rm, ma, dec, decb):
    """STDC_droplet_alpha (decoders.py:510-534): effective lengths n_z + alpha (n_x + n_y) of the distinct chains a Chain_alpha
    visits (`update_chain(5)` per step), in the order found; and STDC_Nall_n_alpha's estimate (:537-581, list form)."""
    rng = np.random.default_rng(1111)
    out = {}
    cases = []
    mods = (xm, rm)
    codes = {"xzzx": xm.xzzx_code, "rot": rm.RotSurCode}
    for i, (name, L, pzt, alpha, steps, perr) in enumerate([("xzzx", 3, 0.2, 2.0, 200, 0.3), ("xzzx", 5, 0.1, 1.7, 200, 0.15),
                                                           ("rot", 5, 0.15, 1.0, 200, 0.15), ("rot", 7, 0.12, 3.1, 150, 0.1)]):
        m = rand_matrix2(rng, L, perr)
        seed = 11000 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        ch = ma.Chain_alpha(np.float64(pzt), alpha, code)
        s = Stream(seed); install(s, *mods)
        seen = dec.STDC_droplet_alpha(ch, steps, alpha)
        restore(*mods)
        tag = f"adrop{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_eff"] = np.array(list(seen.values()), dtype=np.float64)
        out[f"{tag}_final"] = ch.code.qubit_matrix.astype(np.uint8)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, pzt, alpha, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    for i, (name, L, pzs, alpha, pzt, steps, perr) in enumerate([("xzzx", 3, 0.25, 2.0, 0.1, 300, 0.2), ("rot", 5, 0.2, 1.5, 0.08, 250, 0.12),
                                                                ("xzzx", 5, 0.15, 3.0, 0.15, 250, 0.12)]):
        m = rand_matrix2(rng, L, perr)
        seed = 11100 + i
        inits = []
        for op in range(4):
            c = codes[name](L); c.qubit_matrix = m.copy(); c.qubit_matrix = c.apply_logical(op)[0]
            inits.append(c)
        s = Stream(seed); install(s, *mods)
        dist = dec.STDC_Nall_n_alpha(inits, pz_tilde_sampling=np.float64(pzs), alpha=alpha, pz_tilde=pzt, steps=steps)
        restore(*mods)
        tag = f"nall{i}"
        out[f"{tag}_classes"] = np.array([c.qubit_matrix for c in inits], dtype=np.uint8)
        out[f"{tag}_dist"] = np.asarray(dist, dtype=np.float64)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, pzs, alpha, pzt, steps, seed, s.n], dtype=np.float64)
        cases.append(tag)
    # PTEQ_alpha_with_shortest (decoders_biasednoise.py:93-172): PTEQ_alpha plus the shortest-chain statistics of the bottom slot
    for i, (name, L, pzt, alpha, Nc, steps, tops_burn, conv, perr, SEQ, TOPS, eps) in enumerate([
            ("xzzx", 3, 0.2, 2.0, 3, 300, 2, None, 0.3, 2, 10, 0.1), ("xzzx", 5, 0.1, 1.7, 5, 200, 0, None, 0.15, 2, 10, 0.1),
            ("rot", 5, 0.15, 1.3, 5, 200, 1, None, 0.15, 2, 10, 0.1), ("xzzx", 3, 0.3, 2.0, 3, 4000, 2, "error_based", 0.3, 2, 10, 0.6),
            ("rot", 3, 0.25, 1.0, 3, 4000, 2, "error_based", 0.3, 1, 5, 0.8)]):
        m = rand_matrix2(rng, L, perr)
        seed = 11200 + i
        code = codes[name](L); code.qubit_matrix = m.copy()
        s = Stream(seed); install(s, *mods)
        res = decb.PTEQ_alpha_with_shortest(code, np.float64(pzt), alpha=alpha, Nc=Nc, SEQ=SEQ, TOPS=TOPS, tops_burn=tops_burn, eps=eps,
                                            steps=steps, iters=10, conv_criteria=conv)
        restore(*mods)
        tag = f"short{i}"
        out[f"{tag}_init"] = m
        out[f"{tag}_percent"] = np.asarray(res[0], dtype=np.uint8)
        out[f"{tag}_eqdistr"] = np.asarray(res[1], dtype=np.float64)
        out[f"{tag}_shortn"] = np.asarray(res[2], dtype=np.float64)
        out[f"{tag}_par"] = np.array([0 if name == "xzzx" else 1, L, pzt, alpha, Nc, steps, tops_burn, 0 if conv is None else 1, SEQ, TOPS, eps,
                                     seed, s.n], dtype=np.float64)
        cases.append(tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "f_nalpha.npz"), **out)
    print("f_nalpha.npz", cases, {c: int(out[c + "_par"][-1]) for c in cases})


def _f3_worker(args):
    (L, p, Nc, iters, steps, burn, m, seed) = args
    tm, mc, dec = import_reference()
    random.seed(seed)
    code = tm.Toric_code(L); code.qubit_matrix = m.copy()
    ld = mc.Ladder(p, code, Nc, 0.5)
    hist = np.zeros(16, dtype=np.int64)
    nerr = np.zeros(Nc)
    for t in range(steps):
        ld.step(iters)
        if t >= burn:
            hist[ld.chains[0].code.define_equivalence_class()] += 1
            nerr += [c.code.count_errors() for c in ld.chains]
    return hist, nerr / (steps - burn), ld.tops0


def gen_f3(tm):
    """Raw class counts of the bottom chain after a fixed burn-in (the 10-line
    histogram loop around Ladder.step, decoders.py:55-68 with a fixed burn-in),
    R replicas per syndrome, plus per-rung mean error counts (F5)."""
    import multiprocessing as mp
    rng = np.random.default_rng(11)
    out = {}
    configs = [("L5", 5, 0.10, 5, 10, 6000, 1000, 3, 16), ("L3", 3, 0.10, 3, 10, 6000, 1000, 2, 16)]
    with mp.get_context("spawn").Pool(8) as pool:
        for name, L, p, Nc, iters, steps, burn, nsyn, R in configs:
            ms = [rand_matrix(rng, L, p) for _ in range(nsyn)]
            jobs = [(L, p, Nc, iters, steps, burn, ms[s], 5000 + 100 * s + r)
                    for s in range(nsyn) for r in range(R)]
            res = pool.map(_f3_worker, jobs)
            out[f"{name}_init"] = np.array(ms, dtype=np.uint8)
            out[f"{name}_hist"] = np.array([h for h, _, _ in res]).reshape(nsyn, R, 16)
            out[f"{name}_nerr"] = np.array([n for _, n, _ in res]).reshape(nsyn, R, Nc)
            out[f"{name}_tops0"] = np.array([t for _, _, t in res]).reshape(nsyn, R)
            out[f"{name}_par"] = np.array([L, p, Nc, iters, steps, burn], dtype=np.float64)
            print(name, "done")
    np.savez_compressed(os.path.join(HERE, "f3_toric.npz"), **out)


# --------------------------------------------------------------------------- F5
def _f5_worker(args):
    """One replica of a ladder run with the equilibrium observables SURVEY.md 8c lists as F5: class counts of the bottom
    chain, time-averaged count_errors per rung and the outcome of every swap test per adjacent pair (Ladder.r_flip wrapped;
    the sweep visits i = Nc-2 ... 0, src/mcmc.py:96), all after discarding the first `burn` ladder steps."""
    (kind, L, p, eta, Nc, iters, steps, burn, m, seed) = args
    tm, mc, dec = import_reference()
    random.seed(seed)
    if kind == "toric":
        code = tm.Toric_code(L)
    else:
        xm, rm, mb, decb = import_reference_surf()
        code = (xm.xzzx_code if kind.startswith("xzzx") else rm.RotSurCode)(L)
    code.qubit_matrix = m.copy()
    ld = mb.Ladder_biased(p, code, eta, Nc, 0.5) if kind == "xzzx_biased" else mc.Ladder(p, code, Nc, 0.5)
    att = np.zeros(Nc - 1, dtype=np.int64); acc = np.zeros(Nc - 1, dtype=np.int64)
    rec = [False]
    inner = ld.r_flip

    def logged(i):
        r = bool(inner(i))
        if rec[0]:
            att[i] += 1; acc[i] += r
        return r
    ld.r_flip = logged
    ncls = 16 if kind == "toric" else 4
    hist = np.zeros(ncls, dtype=np.int64)
    nerr = np.zeros(Nc)
    for t in range(steps):
        rec[0] = t >= burn
        ld.step(iters)
        if t >= burn:
            hist[ld.chains[0].code.define_equivalence_class()] += 1
            nerr += [c.code.count_errors() for c in ld.chains]
    return hist, nerr / (steps - burn), ld.tops0, att, acc


def gen_f5(only=None):
    """Statistical fixtures where the benchmark lives (SURVEY.md 8c F3/F5, 8d "mixing calibration"): toric L=9 p=0.15 Nc=8,
    rotated L=5/7 p=0.17, Ladder_biased on xzzx L=5/7 eta=100.  R replicas per syndrome; per replica the class counts,
    the per-rung mean error count and the per-pair swap acceptances after a 20 % burn-in."""
    import multiprocessing as mp
    rng = np.random.default_rng(23)
    path = os.path.join(HERE, "f5_stats.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    #           name           kind          L   p     eta  Nc iters steps  burn nsyn R
    configs = [("toric_L9", "toric", 9, 0.15, 0, 8, 10, 20000, 4000, 3, 16),
               ("rot_L5", "rot", 5, 0.17, 0, 5, 10, 20000, 4000, 3, 16),
               ("rot_L7", "rot", 7, 0.17, 0, 7, 10, 20000, 4000, 3, 16),
               ("xzzxb_L5", "xzzx_biased", 5, 0.15, 100, 5, 10, 20000, 4000, 3, 16),
               ("xzzxb_L7", "xzzx_biased", 7, 0.15, 100, 7, 10, 20000, 4000, 3, 16)]
    with mp.get_context("spawn").Pool(8) as pool:
        for name, kind, L, p, eta, Nc, iters, steps, burn, nsyn, R in configs:
            if kind == "toric":
                ms = [rand_matrix(rng, L, p) for _ in range(nsyn)]
            elif kind == "rot":
                ms = [rand_matrix2(rng, L, p) for _ in range(nsyn)]
            else:       # generate_data.py:78-83: p_z = p eta/(eta+1), p_x = p_y = p/(2(eta+1))
                ms = []
                for _ in range(nsyn):
                    r = rng.random((L, L)); pz = p * eta / (eta + 1); px = p / (2 * (eta + 1))
                    m = np.zeros((L, L), dtype=np.uint8)
                    m[r < pz] = 3; m[(r > pz) & (r < pz + px)] = 1; m[(r > pz + px) & (r < pz + 2 * px)] = 2
                    ms.append(m)
            if only and name not in only:
                continue
            jobs = [(kind, L, p, eta, Nc, iters, steps, burn, ms[s], 7000 + 100 * s + r) for s in range(nsyn) for r in range(R)]
            res = pool.map(_f5_worker, jobs)
            ncls = 16 if kind == "toric" else 4
            out[f"{name}_init"] = np.array(ms, dtype=np.uint8)
            out[f"{name}_hist"] = np.array([x[0] for x in res]).reshape(nsyn, R, ncls)
            out[f"{name}_nerr"] = np.array([x[1] for x in res]).reshape(nsyn, R, Nc)
            out[f"{name}_tops0"] = np.array([x[2] for x in res]).reshape(nsyn, R)
            out[f"{name}_swap_att"] = np.array([x[3] for x in res]).reshape(nsyn, R, Nc - 1)
            out[f"{name}_swap_acc"] = np.array([x[4] for x in res]).reshape(nsyn, R, Nc - 1)
            out[f"{name}_par"] = np.array([L, p, eta, Nc, iters, steps, burn], dtype=np.float64)
            print(name, "done: <n> per rung", out[f"{name}_nerr"].mean(axis=(0, 1)).round(1),
                  "swap acceptance", (out[f"{name}_swap_acc"].sum(axis=(0, 1)) / out[f"{name}_swap_att"].sum(axis=(0, 1))).round(3), flush=True)
            np.savez_compressed(path, **out)


# --------------------------------------------------------------------------- F4
def gen_f4(tm, mc):
    random.seed(1); np.random.seed(1)
    code = tm.Toric_code(5)
    code.generate_random_error(0.10)
    init = code.qubit_matrix.copy()
    ch = mc.Chain(0.10, code)
    s = Stream(1); install(s, tm)       # random.Random(1) == the stream after random.seed(1)
    ch.update_chain(10000)
    restore(tm)
    np.savez_compressed(os.path.join(HERE, "f4_config1.npz"), init=init,
                        final=ch.code.qubit_matrix.astype(np.uint8),
                        count=np.int64(ch.code.count_errors()),
                        cls=np.int64(ch.code.define_equivalence_class()),
                        draws=np.int64(s.n), defects=np.asarray(code.defect_matrix, dtype=np.uint8))
    print("f4_config1.npz count", ch.code.count_errors(), "class", ch.code.define_equivalence_class())


# ---- API surface: what the reference's callers touch ---------------------------------------------------------------------------
class _Rec:
    """recording proxy around one of the reference's own objects: every attribute its CALLER (decoders.py, decoders_biasednoise.py,
    generate_data.py) reads, sets or calls goes into `log` as (class, attribute, kind, positional arguments, keywords, result)."""
    log = set()
    classes = ()

    def __init__(self, obj):
        object.__setattr__(self, "_o", obj)

    @staticmethod
    def describe(v):
        if isinstance(v, _Rec):
            v = object.__getattribute__(v, "_o")
        if isinstance(v, np.ndarray):
            return "ndarray%d:%s" % (v.ndim, v.dtype)
        if isinstance(v, tuple):
            return "tuple(" + ",".join(_Rec.describe(x) for x in v) + ")"
        if isinstance(v, (list,)):
            return "list"
        if isinstance(v, (bool, np.bool_)):
            return "bool"
        if isinstance(v, (int, np.integer)):
            return "int"
        if isinstance(v, (float, np.floating)):
            return "float"
        return type(v).__name__

    @staticmethod
    def wrap(v):
        if isinstance(v, _Rec.classes):
            return _Rec(v)
        if isinstance(v, list) and v and isinstance(v[0], _Rec.classes):
            return [_Rec(x) for x in v]
        return v

    @staticmethod
    def unwrap(v):
        if isinstance(v, _Rec):
            return object.__getattribute__(v, "_o")
        if isinstance(v, (list, tuple)):
            return type(v)(_Rec.unwrap(x) for x in v)
        return v

    def __getattr__(self, name):
        o = object.__getattribute__(self, "_o")
        v = getattr(o, name)
        cls = type(o).__name__
        if callable(v) and not isinstance(v, type):
            def call(*a, **k):
                r = v(*_Rec.unwrap(a), **{kk: _Rec.unwrap(vv) for kk, vv in k.items()})
                _Rec.log.add((cls, name, "call", len(a), ",".join(sorted(k)), _Rec.describe(r)))
                return _Rec.wrap(r)
            return call
        _Rec.log.add((cls, name, "get", 0, "", _Rec.describe(v)))
        return _Rec.wrap(v)

    def __setattr__(self, name, val):
        o = object.__getattribute__(self, "_o")
        _Rec.log.add((type(o).__name__, name, "set", 0, "", _Rec.describe(val)))
        setattr(o, name, _Rec.unwrap(val))

    def __deepcopy__(self, memo):
        import copy
        return _Rec(copy.deepcopy(object.__getattribute__(self, "_o"), memo))


class _SerialPool:
    """multiprocessing.Pool stand-in of the recording run: same calls, this process (the workers would take the proxies' logs with them)"""
    def __init__(self, *a, **k): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def starmap(self, fn, args): return [fn(*x) for x in args]
    def map(self, fn, args): return [fn(x) for x in args]


def gen_api(tm, mc, dec):
    """f_api_surface.npz: the names, call shapes and result kinds the reference's decoders use on the sampler path's classes -- recorded
    by running decoders.PTEQ / PTDC / PTRC / STDC / STRC / STDC_general_noise(_shortest) / STDC_Nall_n_alpha / single_temp and
    decoders_biasednoise.PTEQ_biased / PTEQ_alpha / PTEQ_alpha_with_shortest for a few steps on recording proxies of the
    reference's own objects -- plus the names those modules take from `src.*` by import.  tests/test_api_surface.py holds the
    qecmc mirrors to it."""
    import inspect
    import src.planar_model as pm
    import src.xzzx_model as xm
    import src.rotated_surface_model as rm
    import src.mcmc_alpha as ma
    import src.mcmc_biased as mb
    import decoders_biasednoise as decb
    _Rec.log = set()
    _Rec.classes = (mc.Chain, mc.Ladder, mc.Chain_xyz, ma.Chain_alpha, ma.Ladder_alpha, mb.Chain_biased, mb.Ladder_biased,
                    tm.Toric_code, pm.Planar_code, xm.xzzx_code, rm.RotSurCode)
    ctor = set()

    def factory(real):
        def make(*a, **k):
            ctor.add((real.__name__, len(a), ",".join(sorted(k))))
            return _Rec(real(*_Rec.unwrap(a), **{kk: _Rec.unwrap(vv) for kk, vv in k.items()}))
        make.__name__ = real.__name__
        return make
    saved = []
    for mod in (dec, decb):
        for name, val in list(vars(mod).items()):
            if isinstance(val, type) and val in _Rec.classes:
                saved.append((mod, name, val))
                setattr(mod, name, factory(val))
        if hasattr(mod, "Pool"):
            saved.append((mod, "Pool", mod.Pool))
            mod.Pool = _SerialPool
    rng = np.random.default_rng(4242)
    random.seed(77); np.random.seed(77)
    try:
        def toric(L=3, p=0.15):
            c = tm.Toric_code(L); c.qubit_matrix = rand_matrix(rng, L, p); return _Rec(c)
        def planar(L=3, p=0.15):
            c = pm.Planar_code(L); c.qubit_matrix = rand_planar(rng, L, p); return _Rec(c)
        def surf(cls, L=3, p=0.2):
            c = cls(L); c.qubit_matrix = rand_matrix2(rng, L, p); return _Rec(c)
        def class_list(make, n):
            out = []
            base = make()
            for eq in range(n):
                c = _Rec(__import__("copy").deepcopy(_Rec.unwrap(base)))
                out.append(c)
            return out
        dec.PTEQ(toric(), 0.1, Nc=3, steps=60, conv_criteria=None)
        dec.PTEQ(toric(), 0.1, Nc=3, steps=400, TOPS=2, tops_burn=1, eps=5.0)
        dec.PTEQ(planar(), 0.1, Nc=3, steps=60, conv_criteria=None)
        dec.PTEQ(surf(xm.xzzx_code), 0.1, Nc=3, steps=60, conv_criteria=None)
        dec.single_temp(toric(), 0.1, 20)
        dec.PTDC(toric(), 0.1, droplets=1, Nc=2, steps=40)
        def toric_classes():
            base = _Rec.unwrap(toric())
            out = []
            for eq in range(16):
                c = tm.Toric_code(3); c.qubit_matrix = base.to_class(eq); out.append(_Rec(c))
            return out
        def planar_classes():
            m = rand_planar(rng, 3, 0.15)
            out = []
            for op in range(4):
                c = pm.Planar_code(3); c.qubit_matrix, _ = pm._apply_logical(m.copy(), op, 0, 0); out.append(c)
            out.sort(key=lambda c: c.define_equivalence_class())
            return [_Rec(c) for c in out]
        dec.PTRC(toric_classes(), 0.1, droplets=1, Nc=2, steps=40)
        dec.STDC(toric(), 0.1, droplets=1, steps=40)
        dec.STDC(planar_classes(), 0.1, droplets=1, steps=40)
        dec.STRC(planar_classes(), 0.1, droplets=1, steps=40)
        p_xyz = np.array([0.05, 0.03, 0.04])
        inits = planar_classes()
        dec.STDC_general_noise(inits, p_xyz, p_sampling=None, droplets=1, steps=40)
        dec.STDC_general_noise(inits, p_xyz, p_sampling=np.array([0.08, 0.06, 0.07]), droplets=1, steps=40)
        dec.STDC_general_noise_shortest(inits, p_xyz, p_sampling=None, droplets=1, steps=40)
        dec.STDC_Nall_n_alpha([surf(xm.xzzx_code) for _ in range(4)], pz_tilde_sampling=np.float64(0.3), alpha=2.0, pz_tilde=0.2, steps=40)
        decb.PTEQ_biased(surf(xm.xzzx_code), 0.1, eta=10, Nc=3, steps=60, conv_criteria=None)
        decb.PTEQ_biased(surf(rm.RotSurCode), 0.1, eta=10, Nc=3, steps=300, TOPS=2, tops_burn=1, eps=5.0)
        decb.PTEQ_alpha(surf(xm.xzzx_code), 0.1, alpha=2.0, Nc=3, steps=60, conv_criteria=None)
        decb.PTEQ_alpha(surf(rm.RotSurCode), 0.1, alpha=2.0, Nc=3, steps=300, TOPS=2, tops_burn=1, eps=5.0)
        decb.PTEQ_alpha_with_shortest(surf(xm.xzzx_code), 0.1, alpha=2.0, Nc=3, steps=60, conv_criteria=None)
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)
    # what the caller modules take from src.* by import and actually use (co_names of their functions)
    imports = set()
    for mod in (dec, decb):
        used = set()
        for fn in vars(mod).values():
            if inspect.isfunction(fn) and fn.__module__ == mod.__name__:
                used |= set(fn.__code__.co_names)
        for name, val in vars(mod).items():
            m = getattr(val, "__module__", "") or ""
            if name in used and m.startswith("src.") and not m.endswith("mwpm"):
                imports.add((mod.__name__, m, name))
    # generate_data.py (pandas, MWPM: not run here) as text: the attributes it reads / calls on its code object `init_code`
    import ast
    gd = set()
    for node in ast.walk(ast.parse(open(os.path.join(REF, "generate_data.py")).read())):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "init_code":
            gd.add(node.attr)
    recs = sorted("|".join(str(x) for x in r) for r in _Rec.log)
    np.savez(os.path.join(HERE, "f_api_surface.npz"), records=np.array(recs), constructors=np.array(sorted("|".join(str(x) for x in c) for c in ctor)),
             imports=np.array(sorted("|".join(i) for i in imports)), generate_data_code_attrs=np.array(sorted(gd)))
    print("f_api_surface.npz: %d attribute records, %d constructor shapes, %d imported names" % (len(recs), len(ctor), len(imports)))


def main():
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="f1,f2,f3,f4,f1s,f2s,f2a,fp,fd,fc,fg,fa,f5")
    only = set(ap.parse_args().only.split(","))
    tm, mc, dec = import_reference()
    if "f1" in only: gen_f1(tm)
    if "f2" in only: gen_f2(tm, mc, dec)
    if "f4" in only: gen_f4(tm, mc)
    if "api" in only: gen_api(tm, mc, dec)
    if "fp" in only or "fd" in only or "fc" in only or "fg" in only:
        import src.planar_model as pm
        if "fc" in only: gen_convmult(tm, pm, mc, dec)
        if "fg" in only: gen_xyz(tm, pm, mc, dec)
        if "fp" in only: gen_planar(pm, mc, dec)
        if "fd" in only: gen_ptdc(tm, pm, mc, dec)
    if "f1s" in only or "f2s" in only or "f2a" in only or "fa" in only:
        xm, rm, mb, decb = import_reference_surf()
        if "fa" in only:
            import src.mcmc_alpha as ma
            gen_nalpha(xm, rm, ma, dec, decb)
        if "f2a" in only:
            import src.mcmc_alpha as ma
            gen_f2_alpha(xm, rm, ma, decb)
        if "f1s" in only: gen_f1_surf(xm, rm)
        if "f2s" in only: gen_f2_surf(xm, rm, mc, mb, dec, decb)
    if "f3" in only: gen_f3(tm)
    if "f5" in only: gen_f5()


if __name__ == "__main__":
    main()
