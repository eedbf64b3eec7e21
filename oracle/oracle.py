"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.  See oracle/qecmc_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("qecmc_oracle.c", "qecmc_oracle_surf.c", "qecmc_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


class _Rng(C.Structure):
    _fields_ = [("mode", C.c_int), ("stream", C.POINTER(C.c_double)), ("pos", C.c_uint64),
                ("len", C.c_uint64), ("consumed", C.c_uint64), ("seed", C.c_uint64),
                ("syndrome", C.c_uint32), ("c_stream", C.c_uint32), ("c_sub", C.c_uint32),
                ("c_k", C.c_uint64), ("c_valid", C.c_int), ("c_w", C.c_uint32 * 4),
                ("c2_stream", C.c_uint32), ("c2_sub", C.c_uint32), ("c2_k", C.c_uint64), ("c2_valid", C.c_int), ("c2_w", C.c_uint32 * 4),
                ("wave_override", C.c_int), ("wave_group", C.c_uint32), ("wave_t0", C.c_uint64)]


class Model(C.Structure):
    _fields_ = [("code", C.c_int), ("L", C.c_int), ("noise", C.c_int), ("eta", C.c_double), ("alpha", C.c_double),
                ("det_pow", C.c_int), ("scan", C.c_int), ("pxyz", C.c_double * 3)]


TORIC, XZZX, ROTATED, PLANAR = 0, 1, 2, 3
DEPOLARIZING, BIASED, ALPHA, XYZ = 0, 1, 2, 3


class _Ladder(C.Structure):
    _fields_ = [("model", Model), ("L", C.c_int), ("Nc", C.c_int), ("nq", C.c_int), ("p_logical", C.c_double),
                ("p_ladder", C.POINTER(C.c_double)), ("p_diff", C.POINTER(C.c_double)),
                ("states", C.POINTER(C.c_uint8)), ("flags", C.POINTER(C.c_uint8)), ("n_eff", C.POINTER(C.c_double)), ("n_eff_cnt", C.POINTER(C.c_uint32)),
                ("tops0", C.c_uint64), ("step_index", C.c_uint64), ("scratch", C.POINTER(C.c_uint8)),
                ("swap_acc", C.POINTER(C.c_uint64)), ("nerr_sum", C.POINTER(C.c_uint64))]


class PteqResult(C.Structure):
    _fields_ = [("counts", C.c_uint32 * 16), ("samples", C.c_uint64), ("tops0", C.c_uint64),
                ("steps_done", C.c_uint64), ("converged", C.c_int), ("percent", C.c_uint8 * 16)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.environ.get("QECMC_ORACLE_LIB") or build())   # (override: the sanitizer build, oracle/Makefile `asan`)
        u8p = C.POINTER(C.c_uint8)
        _LIB.orc_toric_apply_stabilizer.argtypes = [C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int]
        _LIB.orc_toric_apply_stabilizer.restype = C.c_int
        _LIB.orc_toric_apply_logical.argtypes = [C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB.orc_toric_apply_logical.restype = C.c_int
        _LIB.orc_count_errors.argtypes = [C.c_size_t, u8p]
        _LIB.orc_count_errors.restype = C.c_int64
        _LIB.orc_toric_eq_class.argtypes = [C.c_int, u8p]
        _LIB.orc_toric_eq_class.restype = C.c_int
        _LIB.orc_toric_to_class.argtypes = [C.c_int, u8p, u8p, C.c_int]
        _LIB.orc_toric_to_class.restype = None
        _LIB.orc_toric_syndrome.argtypes = [C.c_int, u8p, u8p]
        _LIB.orc_toric_syndrome.restype = None
        _LIB.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _LIB.orc_philox4x32_10.restype = None
        _LIB.orc_rng_init_stream.argtypes = [C.POINTER(_Rng), C.POINTER(C.c_double), C.c_uint64]
        _LIB.orc_rng_init_philox.argtypes = [C.POINTER(_Rng), C.c_uint64, C.c_uint32]
        _LIB.orc_toric_chain_update.argtypes = [C.c_int, u8p, C.c_double, C.c_double, C.c_uint64,
                                                C.POINTER(_Rng), C.c_uint32, C.c_uint64, u8p]
        _LIB.orc_toric_chain_update.restype = None
        _LIB.orc_toric_ladder_new.argtypes = [C.c_int, u8p, C.c_double, C.c_int, C.c_double]
        _LIB.orc_toric_ladder_new.restype = C.POINTER(_Ladder)
        _LIB.orc_ladder_free.argtypes = [C.POINTER(_Ladder)]
        _LIB.orc_toric_ladder_step.argtypes = [C.POINTER(_Ladder), C.c_uint64, C.POINTER(_Rng)]
        _LIB.orc_toric_pteq.argtypes = [C.c_int, u8p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(_Rng),
                                        C.POINTER(PteqResult), u8p]
        _LIB.orc_toric_pteq.restype = None
        _LIB.orc_toric_pteq_batch.argtypes = [C.c_int, u8p, C.c_uint64, C.c_uint32, C.c_double, C.c_int,
                                              C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64), u8p]
        _LIB.orc_toric_pteq_batch.restype = None
        _LIB.orc_toric_pteq_batch_conv.argtypes = [C.c_int, u8p, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_int,
                                                   C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), u8p, u8p]
        _LIB.orc_toric_pteq_batch_conv.restype = None
        mp = C.POINTER(Model)
        _LIB.orc_surf_apply_stabilizer.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int]
        _LIB.orc_surf_apply_stabilizer.restype = C.c_int
        _LIB.orc_surf_apply_logical.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int]
        _LIB.orc_surf_apply_logical.restype = C.c_int
        _LIB.orc_surf_eq_class.argtypes = [C.c_int, C.c_int, u8p]
        _LIB.orc_surf_eq_class.restype = C.c_int
        _LIB.orc_surf_syndrome.argtypes = [C.c_int, C.c_int, u8p, u8p]
        _LIB.orc_surf_syndrome.restype = None
        _LIB.orc_planar_syndrome.argtypes = [C.c_int, u8p, u8p, u8p]
        _LIB.orc_planar_syndrome.restype = None
        _LIB.orc_surf_ngen.argtypes = [C.c_int, C.c_int]; _LIB.orc_surf_ngen.restype = C.c_int
        _LIB.orc_colour_phases.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]; _LIB.orc_colour_phases.restype = C.c_int
        _LIB.orc_surf_gen_rco.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _LIB.orc_surf_gen_rco.restype = None
        _LIB.orc_chain_update.argtypes = [mp, u8p, C.c_double, C.c_double, C.c_uint64, C.POINTER(_Rng), C.c_uint32,
                                          C.c_uint64, u8p]
        _LIB.orc_chain_update.restype = None
        _LIB.orc_chain_update_alpha.argtypes = [mp, u8p, C.c_double, C.c_double, C.c_uint64, C.POINTER(_Rng), C.c_uint32,
                                                C.c_uint64, u8p, C.POINTER(C.c_double)]
        _LIB.orc_chain_update_alpha.restype = C.c_int
        _LIB.orc_det_exp.argtypes = [C.c_double]; _LIB.orc_det_exp.restype = C.c_double
        u64p = C.POINTER(C.c_uint64); u32p_ = C.POINTER(C.c_uint32)
        _LIB.orc_ptdc_droplet.argtypes = [mp, u8p, C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(_Rng), u64p, C.c_uint64, u32p_,
                                          C.c_int, u32p_, C.c_double, u64p, C.POINTER(_XyzSink)]
        _LIB.orc_ptdc_droplet.restype = None
        _LIB.orc_ptdc_batch.argtypes = [mp, u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_uint64,
                                        C.c_uint64, C.c_uint64, C.c_int, u32p_, C.c_int, u32p_, C.c_double, u32p_]
        _LIB.orc_ptdc_batch.restype = None
        _LIB.orc_state_key.argtypes = [u8p, C.c_size_t]; _LIB.orc_state_key.restype = C.c_uint64
        _LIB.orc_ladder_new.argtypes = [mp, u8p, C.c_double, C.c_int, C.c_double]
        _LIB.orc_ladder_new.restype = C.POINTER(_Ladder)
        _LIB.orc_ladder_step.argtypes = [C.POINTER(_Ladder), C.c_uint64, C.POINTER(_Rng)]
        _LIB.orc_pteq.argtypes = [mp, u8p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
                                  C.c_uint64, C.c_int, C.POINTER(_Rng), C.POINTER(PteqResult), u8p]
        _LIB.orc_pteq.restype = None
        _LIB.orc_pteq_batch.argtypes = [mp, u8p, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_int,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), u8p, u8p]
        _LIB.orc_pteq_batch.restype = None
        _LIB.orc_pteq_wave_queue.argtypes = [mp, u8p, C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint64), u8p]
        _LIB.orc_pteq_wave_queue.restype = None
    return _LIB


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _m(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class Rng:
    """Uniform source: Rng.stream(array) or Rng.philox(seed, syndrome)."""

    def __init__(self):
        self.c = _Rng()
        self._keep = None

    @classmethod
    def stream(cls, values):
        r = cls()
        r._keep = np.ascontiguousarray(values, dtype=np.float64)
        lib().orc_rng_init_stream(C.byref(r.c), r._keep.ctypes.data_as(C.POINTER(C.c_double)), r._keep.size)
        return r

    @classmethod
    def philox(cls, seed, syndrome=0):
        r = cls()
        lib().orc_rng_init_philox(C.byref(r.c), seed, syndrome)
        return r

    @property
    def consumed(self):
        return int(self.c.consumed)


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


# ---- toric stencils ---------------------------------------------------------
def toric_apply_stabilizer(m, row, col, op):
    m = _m(m); L = m.shape[1]; out = np.empty_like(m)
    dE = lib().orc_toric_apply_stabilizer(L, _u8(m), _u8(out), row, col, op)
    return out, dE


def toric_apply_logical(m, op, layer, xpos=0, zpos=0):
    m = _m(m); L = m.shape[1]; out = np.empty_like(m)
    dE = lib().orc_toric_apply_logical(L, _u8(m), _u8(out), op, layer, xpos, zpos)
    return out, dE


def count_errors(m):
    m = _m(m)
    return int(lib().orc_count_errors(m.size, _u8(m)))


def toric_eq_class(m):
    m = _m(m)
    return int(lib().orc_toric_eq_class(m.shape[1], _u8(m)))


def toric_to_class(m, eq):
    m = _m(m); out = np.empty_like(m)
    lib().orc_toric_to_class(m.shape[1], _u8(m), _u8(out), eq)
    return out


def toric_syndrome(m):
    m = _m(m); out = np.empty_like(m)
    lib().orc_toric_syndrome(m.shape[1], _u8(m), _u8(out))
    return out


# ---- chain / ladder / PTEQ --------------------------------------------------
def toric_chain_update(m, p, p_logical, iters, rng, slot=0, k0=0):
    m = _m(m).copy(); L = m.shape[1]
    scratch = np.empty_like(m)
    lib().orc_toric_chain_update(L, _u8(m), p, p_logical, iters, C.byref(rng.c), slot, k0, _u8(scratch))
    return m


class ToricLadder:
    def __init__(self, init, p_bottom, Nc, p_logical=0.0):
        init = _m(init)
        self.L = init.shape[1]; self.Nc = Nc; self.nq = init.size
        self._p = lib().orc_toric_ladder_new(self.L, _u8(init), p_bottom, Nc, p_logical)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_ladder_free(self._p)
            self._p = None

    def step(self, iters, rng):
        lib().orc_toric_ladder_step(self._p, iters, C.byref(rng.c))

    @property
    def states(self):
        a = np.ctypeslib.as_array(self._p.contents.states, shape=(self.Nc * self.nq,))
        return a.reshape(self.Nc, 2, self.L, self.L).copy()

    @property
    def flags(self):
        return np.ctypeslib.as_array(self._p.contents.flags, shape=(self.Nc,)).copy()

    @property
    def tops0(self):
        return int(self._p.contents.tops0)

    @property
    def p_ladder(self):
        return np.ctypeslib.as_array(self._p.contents.p_ladder, shape=(self.Nc,)).copy()

    @property
    def p_diff(self):
        return np.ctypeslib.as_array(self._p.contents.p_diff, shape=(self.Nc - 1,)).copy()

    swap_accepts = property(lambda self: Ladder.swap_accepts.fget(self))
    nerr_sums = property(lambda self: Ladder.nerr_sums.fget(self))


def toric_pteq(init, p, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=1000, iters=10,
               conv_criteria=None, rng=None, return_states=False):
    init = _m(init); L = init.shape[1]; Nc = Nc or L
    res = PteqResult()
    fin = np.empty((Nc,) + init.shape, dtype=np.uint8)
    lib().orc_toric_pteq(L, _u8(init), p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters,
                         1 if conv_criteria == "error_based" else 0, C.byref(rng.c), C.byref(res), _u8(fin))
    out = dict(counts=np.array(res.counts[:], dtype=np.uint32), samples=int(res.samples),
               tops0=int(res.tops0), steps_done=int(res.steps_done), converged=bool(res.converged),
               percent=np.array(res.percent[:], dtype=np.uint8))
    if return_states:
        out["states"] = fin
    return out


def toric_pteq_batch(init, p, Nc, steps, iters=10, tops_burn=2, seed=0, first_syndrome=0, n_threads=0,
                     return_states=False, conv_criteria=None, SEQ=2, TOPS=10, eps=0.1):
    init = _m(init); N = init.shape[0]; L = init.shape[2]
    counts = np.zeros((N, 16), dtype=np.uint32)
    samples = np.zeros(N, dtype=np.uint64)
    tops0 = np.zeros(N, dtype=np.uint64)
    steps_done = np.zeros(N, dtype=np.uint64)
    converged = np.zeros(N, dtype=np.uint8)
    fin = np.empty((N, Nc) + init.shape[1:], dtype=np.uint8) if return_states else None
    lib().orc_toric_pteq_batch_conv(L, _u8(init), N, first_syndrome, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters,
                                    1 if conv_criteria == "error_based" else 0, seed, n_threads,
                                    counts.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    samples.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    tops0.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    steps_done.ctypes.data_as(C.POINTER(C.c_uint64)), _u8(converged),
                                    _u8(fin) if fin is not None else None)
    out = dict(counts=counts, samples=samples, tops0=tops0, steps_done=steps_done, converged=converged.astype(bool))
    if return_states:
        out["states"] = fin
    return out


# ---- XZZX / rotated surface code (uint8[L,L]) and the code/noise-generic chain, ladder, PTEQ --------
def _model(code, L, noise=DEPOLARIZING, eta=0.0, scan=0, alpha=0.0, det_pow=0, pxyz=None):
    if pxyz is not None:
        noise = XYZ
    return Model(code, L, noise, float(eta), float(alpha), det_pow, scan, (C.c_double * 3)(*(pxyz if pxyz is not None else (0, 0, 0))))


def _size(code, m):
    return m.shape[-1]


def surf_apply_stabilizer(code, m, row, col, op):
    m = _m(m); out = np.empty_like(m)
    dE = lib().orc_surf_apply_stabilizer(code, m.shape[-1], _u8(m), _u8(out), row, col, op)
    return out, dE


def surf_apply_logical(code, m, op, xpos=0, zpos=0):
    m = _m(m); out = np.empty_like(m)
    dE = lib().orc_surf_apply_logical(code, m.shape[-1], _u8(m), _u8(out), op, xpos, zpos)
    return out, dE


def surf_eq_class(code, m):
    m = _m(m)
    return int(lib().orc_surf_eq_class(code, m.shape[-1], _u8(m)))


def planar_syndrome(m):
    """(vertex_defects bool[L-1, L], plaquette_defects bool[L, L-1]) of Planar_code.syndrom (planar_model.py:134-153)."""
    m = _m(m); L = m.shape[-1]
    v = np.zeros((L - 1, L), dtype=np.uint8); q = np.zeros((L, L - 1), dtype=np.uint8)
    lib().orc_planar_syndrome(L, _u8(m), _u8(v), _u8(q))
    return v.astype(bool), q.astype(bool)


def colour_phases(code, L):
    """scan = 2: int32[n_phases, 64] generator indices of the colour phases (-1 = none), the oracle's own statement of the rule"""
    buf = np.full(64 * 64, -1, dtype=np.int32)
    n = int(lib().orc_colour_phases(code, L, buf.ctypes.data_as(C.POINTER(C.c_int)), buf.size))
    if n < 0 or n > 64:
        raise RuntimeError("colour_phases: %d" % n)
    return buf[:n * 64].reshape(n, 64).copy()


def surf_ngen(code, L):
    return int(lib().orc_surf_ngen(code, L))


def surf_gen_rco(code, L, g):
    r, c, o = C.c_int(), C.c_int(), C.c_int()
    lib().orc_surf_gen_rco(code, L, g, C.byref(r), C.byref(c), C.byref(o))
    return r.value, c.value, o.value


def surf_syndrome(code, m):
    m = _m(m); L = m.shape[0]
    out = np.zeros((L + 1, L + 1), dtype=np.uint8)
    lib().orc_surf_syndrome(code, L, _u8(m), _u8(out))
    return out


def chain_update(code, m, p, p_logical, iters, rng, slot=0, k0=0, noise=DEPOLARIZING, eta=0.0, scan=0, pxyz=None):
    m = _m(m).copy()
    scratch = np.empty_like(m)
    mod = _model(code, _size(code, m), XYZ if pxyz is not None else noise, eta, scan, pxyz=pxyz)
    lib().orc_chain_update(C.byref(mod), _u8(m), p, p_logical, iters, C.byref(rng.c), slot, k0, _u8(scratch))
    return m


def chain_update_alpha(code, m, pz_tilde, alpha, p_logical, iters, rng, slot=0, k0=0):
    """Chain_alpha.update_chain; returns (final matrix, n_eff attribute)."""
    m = _m(m).copy()
    scratch = np.empty_like(m)
    mod = _model(code, _size(code, m), ALPHA, 0.0, 0, alpha)
    n_eff = C.c_double(float(np.sum(m == 3) + alpha * (np.sum(m == 1) + np.sum(m == 2))))
    lib().orc_chain_update_alpha(C.byref(mod), _u8(m), pz_tilde, p_logical, iters, C.byref(rng.c), slot, k0, _u8(scratch),
                                 C.byref(n_eff))
    return m, n_eff.value


class Ladder:
    """Ladder / Ladder_biased / Ladder_alpha of any code model (states in slot order)."""

    def __init__(self, code, init, p_bottom, Nc, p_logical=0.0, noise=DEPOLARIZING, eta=0.0, scan=0, alpha=0.0, det_pow=0):
        init = _m(init)
        self.shape = init.shape
        self.Nc = Nc; self.nq = init.size
        mod = _model(code, _size(code, init), noise, eta, scan, alpha, det_pow)
        self._p = lib().orc_ladder_new(C.byref(mod), _u8(init), p_bottom, Nc, p_logical)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_ladder_free(self._p)
            self._p = None

    def step(self, iters, rng):
        lib().orc_ladder_step(self._p, iters, C.byref(rng.c))

    @property
    def states(self):
        a = np.ctypeslib.as_array(self._p.contents.states, shape=(self.Nc * self.nq,))
        return a.reshape((self.Nc,) + self.shape).copy()

    @property
    def flags(self):
        return np.ctypeslib.as_array(self._p.contents.flags, shape=(self.Nc,)).copy()

    @property
    def tops0(self):
        return int(self._p.contents.tops0)

    @property
    def p_ladder(self):
        return np.ctypeslib.as_array(self._p.contents.p_ladder, shape=(self.Nc,)).copy()

    @property
    def p_diff(self):
        return np.ctypeslib.as_array(self._p.contents.p_diff, shape=(self.Nc - 1,)).copy()

    @property
    def swap_accepts(self):
        """accepted swap tests per rung pair since construction"""
        return np.ctypeslib.as_array(self._p.contents.swap_acc, shape=(max(self.Nc - 1, 1),))[:self.Nc - 1].copy()

    @property
    def nerr_sums(self):
        """sum over the steps of each rung's error count after the step's swaps"""
        return np.ctypeslib.as_array(self._p.contents.nerr_sum, shape=(self.Nc,)).copy()

    @property
    def n_eff(self):
        return np.ctypeslib.as_array(self._p.contents.n_eff, shape=(self.Nc,)).copy()

    @property
    def n_eff_counts(self):
        return np.ctypeslib.as_array(self._p.contents.n_eff_cnt, shape=(self.Nc, 2)).copy()


def pteq(code, init, p, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=1000, iters=10, conv_criteria=None,
         rng=None, noise=DEPOLARIZING, eta=0.0, alpha=0.0, det_pow=0, scan=0):
    init = _m(init); Nc = Nc or _size(code, init)
    mod = _model(code, _size(code, init), noise, eta, scan, alpha, det_pow)
    res = PteqResult()
    fin = np.empty((Nc,) + init.shape, dtype=np.uint8)
    lib().orc_pteq(C.byref(mod), _u8(init), p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters,
                   1 if conv_criteria == "error_based" else 0, C.byref(rng.c), C.byref(res), _u8(fin))
    ncls = 16 if code == TORIC else 4
    return dict(counts=np.array(res.counts[:ncls], dtype=np.uint32), samples=int(res.samples), tops0=int(res.tops0),
                steps_done=int(res.steps_done), converged=bool(res.converged),
                percent=np.array(res.percent[:ncls], dtype=np.uint8), states=fin)


def pteq_batch(code, init, p, Nc, steps, iters=10, tops_burn=2, seed=0, first_syndrome=0, n_threads=0,
               return_states=False, conv_criteria=None, SEQ=2, TOPS=10, eps=0.1, noise=DEPOLARIZING, eta=0.0, scan=0,
               alpha=0.0, det_pow=0):
    init = _m(init); N = init.shape[0]
    mod = _model(code, init.shape[-1], noise, eta, scan, alpha, det_pow)
    counts = np.zeros((N, 16), dtype=np.uint32)
    samples = np.zeros(N, dtype=np.uint64); tops0 = np.zeros(N, dtype=np.uint64)
    steps_done = np.zeros(N, dtype=np.uint64); converged = np.zeros(N, dtype=np.uint8)
    fin = np.empty((N, Nc) + init.shape[1:], dtype=np.uint8) if return_states else None
    lib().orc_pteq_batch(C.byref(mod), _u8(init), N, first_syndrome, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters,
                         1 if conv_criteria == "error_based" else 0, seed, n_threads,
                         counts.ctypes.data_as(C.POINTER(C.c_uint32)), samples.ctypes.data_as(C.POINTER(C.c_uint64)),
                         tops0.ctypes.data_as(C.POINTER(C.c_uint64)), steps_done.ctypes.data_as(C.POINTER(C.c_uint64)),
                         _u8(converged), _u8(fin) if fin is not None else None)
    ncls = 16 if code == TORIC else 4
    out = dict(counts=counts[:, :ncls].copy(), samples=samples, tops0=tops0, steps_done=steps_done,
               converged=converged.astype(bool))
    if return_states:
        out["states"] = fin
    return out


def pteq_wave_queue(code, init, p, Nc, steps, grid, iters=10, tops_burn=2, seed=0, first_syndrome=0, n_threads=0, SEQ=2, TOPS=10, eps=0.1,
                    noise=DEPOLARIZING, alpha=0.0, det_pow=0):
    """scan = 3 with the error_based criterion on a persistent grid of `grid` workgroups: the deterministic work queue of the GPU
    kernel, restated (orc_pteq_wave_queue)"""
    init = _m(init); N = init.shape[0]
    mod = _model(code, init.shape[-1], noise, 0.0, 3, alpha, det_pow)
    counts = np.zeros((N, 16), dtype=np.uint32)
    samples = np.zeros(N, dtype=np.uint64); tops0 = np.zeros(N, dtype=np.uint64)
    steps_done = np.zeros(N, dtype=np.uint64); converged = np.zeros(N, dtype=np.uint8)
    lib().orc_pteq_wave_queue(C.byref(mod), _u8(init), N, first_syndrome, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, seed, grid, n_threads,
                              counts.ctypes.data_as(C.POINTER(C.c_uint32)), samples.ctypes.data_as(C.POINTER(C.c_uint64)),
                              tops0.ctypes.data_as(C.POINTER(C.c_uint64)), steps_done.ctypes.data_as(C.POINTER(C.c_uint64)), _u8(converged))
    ncls = 16 if code == TORIC else 4
    return dict(counts=counts[:, :ncls].copy(), samples=samples, tops0=tops0, steps_done=steps_done, converged=converged.astype(bool))


# ---- unique-chain estimators (decoders.py:138-233) ---------------------------------------------------------------
def _sampling_model(code, L, p_sampling, alpha=None):
    """scalar p_sampling: Chain / Ladder (src/mcmc.py); a (p_x, p_y, p_z) array: Chain_xyz (mcmc.py:106-114), single chains only;
    alpha: Chain_alpha at pz_tilde = p_sampling (src/mcmc_alpha.py; STDC_droplet_alpha, decoders.py:510-534)"""
    if alpha is not None:
        return _model(code, L, noise=ALPHA, alpha=alpha), float(p_sampling)
    if np.ndim(p_sampling) == 0:
        return _model(code, L), float(p_sampling)
    ps = [float(v) for v in p_sampling]
    return _model(code, L, pxyz=ps), float(sum(ps))


class _XyzSink(C.Structure):
    _fields_ = [("out", C.POINTER(C.c_uint32)), ("n", C.c_uint64)]


def unpack_xyz(vals):
    """n_x | n_y << 10 | n_z << 20 words -> int64[k, 3]; the unused 0xFFFFFFFF tail is dropped."""
    v = np.asarray(vals, dtype=np.uint32)
    v = v[v != 0xFFFFFFFF]
    return np.stack([v & 1023, (v >> 10) & 1023, (v >> 20) & 1023], axis=-1).astype(np.int64)


def ptdc_droplet(code, init, p_sampling, Nc, steps, iters=10, rng=None, tab=None, per_rung=False, with_m=False, conv_mult=0.0,
                 with_xyz=False, alpha=None):
    """PTDC_droplet (conv_mult = 0): returns (N(n) uint32[nq+1] of the chains that were new to `tab`, tab).  Nc=1, iters=5 is
    STDC_droplet / STRC_droplet; per_rung=True is PTRC_droplet (N(n) per rung, uint32[Nc, nq+1]); with_m also returns m(n).
    with_xyz: returns (N(n), xyz int64[k, 3], tab) -- (n_x, n_y, n_z) of the k chains new to tab, in the order found
    (STDC_droplet_general_noise's dict values, decoders.py:325-342)."""
    init = _m(init); nq = init.size
    mod, p_sampling = _sampling_model(code, init.shape[-1], p_sampling, alpha)
    nset = Nc if per_rung else 1
    if tab is None:
        cap = 16
        while cap < 2 * steps * (1 if per_rung else Nc):
            cap <<= 1
        tab = np.zeros(cap * nset, dtype=np.uint64)
    hist = np.zeros((nset, nq + 1), dtype=np.uint32)
    mh = np.zeros((nset, nq + 1), dtype=np.uint32) if with_m else None
    u32p = C.POINTER(C.c_uint32)
    xv = np.full(steps * Nc, 0xFFFFFFFF, dtype=np.uint32) if with_xyz else None
    sink = _XyzSink(xv.ctypes.data_as(u32p), 0) if with_xyz else None
    lib().orc_ptdc_droplet(C.byref(mod), _u8(init), p_sampling, Nc, steps, iters, C.byref(rng.c),
                           tab.ctypes.data_as(C.POINTER(C.c_uint64)), tab.size // nset, hist.ctypes.data_as(u32p), int(per_rung),
                           mh.ctypes.data_as(u32p) if with_m else None, float(conv_mult), None, C.byref(sink) if with_xyz else None)
    if not per_rung:
        hist = hist[0]; mh = mh[0] if with_m else None
    if with_xyz:
        return hist, unpack_xyz(xv[:sink.n]), tab
    return (hist, mh, tab) if with_m else (hist, tab)


def ptdc_batch(code, init, p_sampling, Nc, steps, droplets=1, iters=10, seed=0, first_syndrome=0, n_threads=0, per_rung=False,
               with_m=False, conv_mult=0.0, with_xyz=False, alpha=None):
    """init uint8[N, ncls, ...] class representatives (or [N, ncls, droplets, ...]) -> N(n) uint32[N, ncls, nq+1]
    (per_rung: [N, ncls, droplets, Nc, nq+1]); with_m: (N(n), m(n)); with_xyz appends uint32[N, ncls, steps*Nc*droplets],
    the packed (n_x, n_y, n_z) of every distinct chain of the class set (tail 0xFFFFFFFF)."""
    init = _m(init); N, ncls = init.shape[0], init.shape[1]
    nd = 3 if code in (TORIC, PLANAR) else 2
    per_droplet = init.ndim == nd + 3
    nq = int(np.prod(init.shape[-nd:]))
    mod, p_sampling = _sampling_model(code, init.shape[-1], p_sampling, alpha)
    shape = (N, ncls, droplets, Nc, nq + 1) if per_rung else (N, ncls, nq + 1)
    hist = np.zeros(shape, dtype=np.uint32)
    mh = np.zeros(shape, dtype=np.uint32) if with_m else None
    u32p = C.POINTER(C.c_uint32)
    xv = np.zeros((N, ncls, steps * Nc * droplets), dtype=np.uint32) if with_xyz else None
    lib().orc_ptdc_batch(C.byref(mod), _u8(init), N, ncls, droplets, int(per_droplet), first_syndrome, p_sampling, Nc, steps, iters,
                         seed, n_threads, hist.ctypes.data_as(u32p), int(per_rung), mh.ctypes.data_as(u32p) if with_m else None,
                         float(conv_mult), xv.ctypes.data_as(u32p) if with_xyz else None)
    out = (hist, mh) if with_m else (hist,)
    if with_xyz:
        out = out + (xv,)
    return out if len(out) > 1 else out[0]


def ptdc_distribution(hist, p_error):
    """eqdistr of decoders.py:208,229-233: Z_E = sum over unique chains of exp(-beta n), normalised, in percent."""
    beta = -np.log((p_error / 3) / (1 - p_error))
    n = np.arange(hist.shape[-1], dtype=np.float64)
    Z = (hist.astype(np.float64) * np.exp(-beta * n)).sum(axis=-1)
    return Z / Z.sum(axis=-1, keepdims=True) * 100


def generate_syndromes(code, L, N, p_x, p_y, p_z, hide_class=True, seed=0, first_syndrome=0):
    """N error chains + (hide_class) a random logical operator, Philox mode: (init, raw, eq_true)."""
    shape = (N, 2, L, L) if code in (TORIC, PLANAR) else (N, L, L)
    init = np.zeros(shape, dtype=np.uint8); raw = np.zeros(shape, dtype=np.uint8); eq = np.zeros(N, dtype=np.int32)
    lib().orc_generate_syndromes(code, L, C.c_uint64(N), C.c_double(p_x), C.c_double(p_y), C.c_double(p_z), int(bool(hide_class)),
                                 C.c_uint64(seed), C.c_uint32(first_syndrome), _u8(init), _u8(raw), eq.ctypes.data_as(C.POINTER(C.c_int32)))
    return init, raw, eq
