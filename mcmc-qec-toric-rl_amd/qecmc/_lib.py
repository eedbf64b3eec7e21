"""ctypes binding of libqecmc.so (C-ABI: include/qecmc.h).

The library is the product: there is no Python/NumPy compute fallback.  If the
shared object is missing, or no MI355X is visible when a compute entry point
is called, the call raises -- loudly -- instead of degrading.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqecmc.so")
FLAG_NO_PRE, FLAG_NO_DELUT, FLAG_NO_SSW = 2, 4, 8          # qecmc_flag (include/qecmc.h): developer switches between equivalent kernel variants


def dev_flags(switches=0, queue_grid=0):
    """qecmc_params.flags: qecmc_flag switches | the work-queue kernels' persistent grid in workgroups (0: what fits the chip)"""
    return (int(switches) & 0xFFFF) | (int(queue_grid) << 16)


def use_library(path):
    """Load another build of the same ABI instead of the in-tree libqecmc.so (A/B timing of two builds in one process tree:
    `bench.py --library`).  Must be called before the first entry point is used."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise QecmcError("use_library() after the library has been loaded")
    LIB_PATH = os.path.abspath(path)

TORIC, XZZX, ROTATED, PLANAR = 0, 1, 2, 3
SCAN_RANDOM, SCAN_SWEEP, SCAN_COLOUR, SCAN_WAVE = 0, 1, 2, 3
SCANS = {"random": SCAN_RANDOM, "sweep": SCAN_SWEEP, "colour": SCAN_COLOUR, "wave": SCAN_WAVE}
NOISE_DEPOLARIZING, NOISE_BIASED, NOISE_ALPHA = 0, 1, 2
CONV_NONE, CONV_ERROR_BASED = 0, 1
PTDC_INIT_PER_DROPLET, PTDC_SET_PER_RUNG = 1, 2


class QecmcError(RuntimeError):
    pass


class Params(C.Structure):
    """struct qecmc_params (include/qecmc.h)."""
    _fields_ = [("abi_size", C.c_uint32), ("code", C.c_int32), ("L", C.c_int32), ("Nc", C.c_int32),
                ("noise", C.c_int32), ("scan", C.c_int32), ("conv_mode", C.c_int32), ("device", C.c_int32),
                ("iters", C.c_uint64), ("steps", C.c_uint64), ("tops_burn", C.c_int32), ("TOPS", C.c_int32),
                ("SEQ", C.c_int32), ("replicas", C.c_int32), ("eps", C.c_double), ("p", C.c_double),
                ("eta", C.c_double), ("alpha", C.c_double), ("p_logical", C.c_double), ("seed", C.c_uint64),
                ("first_syndrome", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("proposals", C.c_uint64), ("swap_tests", C.c_uint64), ("kernel_ms", C.c_double),
                ("total_ms", C.c_double)]


_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)
_u16p = C.POINTER(C.c_uint16)

# every symbol include/qecmc.h declares, with its signature
SIGNATURES = {
    "qecmc_abi_version": (C.c_int, []),
    "qecmc_last_error": (C.c_char_p, []),
    "qecmc_device_count": (C.c_int, []),
    "qecmc_apply_stabilizer": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, _u8p, _i32p, _i32p, _i32p, _i32p]),
    "qecmc_apply_logical": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, _u8p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "qecmc_count_errors": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, _i64p]),
    "qecmc_eq_class": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, _i32p]),
    "qecmc_to_class": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, _u8p, _i32p]),
    "qecmc_syndrome": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, _u8p]),
    "qecmc_generate_syndromes": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint64,
                                           C.c_uint32, _u8p, _u8p, _i32p]),
    "qecmc_generate_syndromes_dev": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int,
                                               C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qecmc_chain_update": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, C.c_double, C.c_double, C.c_uint64,
                                     C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]),
    "qecmc_chain_update_biased": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, C.c_double, C.c_double, C.c_double,
                                            C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]),
    "qecmc_chain_update_alpha": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, C.c_double, C.c_double, C.c_double,
                                           C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, _u8p]),
    "qecmc_chain_update_xyz": (C.c_int, [C.c_int, C.c_int, C.c_uint64, _u8p, C.POINTER(C.c_double), C.c_uint64, C.c_uint64, C.c_uint32,
                                         C.c_uint32, C.c_uint64]),
    "qecmc_ladder_step": (C.c_int, [C.POINTER(Params), C.c_uint64, _u8p, _u8p, _u32p, C.c_uint64, C.c_uint64,
                                    C.c_uint64, C.c_uint64]),
    "qecmc_ladder_step_alpha": (C.c_int, [C.POINTER(Params), C.c_uint64, _u8p, _u8p, _u32p, _u16p, C.c_uint64,
                                          C.c_uint64, C.c_uint64, C.c_uint64]),
    "qecmc_pteq_batch": (C.c_int, [C.POINTER(Params), _u8p, C.c_uint64, _u32p, _u32p, _u32p, _u32p, _u8p, _u8p,
                                   C.POINTER(Stats)]),
    "qecmc_ptdc_batch": (C.c_int, [C.POINTER(Params), _u8p, C.c_uint64, C.c_int32, C.c_uint32, _u32p, _u32p, C.POINTER(Stats)]),
    "qecmc_ptdc_batch_conv": (C.c_int, [C.POINTER(Params), _u8p, C.c_uint64, C.c_int32, C.c_uint32, C.c_double, _u32p, _u32p, _u32p,
                                        C.POINTER(Stats)]),
    "qecmc_ptdc_batch_xyz": (C.c_int, [C.POINTER(Params), _u8p, C.c_uint64, C.c_int32, C.c_uint32, C.c_double, C.POINTER(C.c_double),
                                       _u32p, _u32p, _u32p, _u32p, _u32p, C.POINTER(Stats)]),
    "qecmc_pteq_batch_stats": (C.c_int, [C.POINTER(Params), _u8p, C.c_uint64, _u32p, _u32p, _u32p, _u32p, _u8p, _u8p, _u32p, _u32p,
                                         C.POINTER(Stats)]),
    "qecmc_plan_set_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "qecmc_pteq_resume_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "qecmc_plan_workspace_bytes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    "qecmc_plan_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "qecmc_plan_destroy": (C.c_int, [C.c_void_p]),
    "qecmc_plan_info": (C.c_int, [C.c_void_p, _u32p, _u32p, _u32p]),
    "qecmc_pteq_launch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheels bundle their own HIP runtime.  A process that loads libqecmc (linked against /opt/rocm) first and
    initialises torch.cuda afterwards would hold two runtimes and torch then finds no device; loading torch's copy first makes
    both sides resolve the same libamdhip64 (what happens anyway when torch is imported before this module is used).
    No torch installed: nothing to do -- the library does not need it."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QecmcError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.qecmc_abi_version() != 4:
            raise QecmcError("libqecmc ABI version mismatch")
        _lib = L
    return _lib


def library_path():
    """the libqecmc.so this process binds (QECMC_LIBRARY or the in-tree build): bench lines carry it, so an A/B names its builds"""
    return os.path.abspath(LIB_PATH)


def check(rc):
    if rc != 0:
        raise QecmcError(f"libqecmc error {rc}: {lib().qecmc_last_error().decode()}")


def device_count():
    return lib().qecmc_device_count()


def u8(a):
    return a.ctypes.data_as(_u8p)


def i32(a):
    return a.ctypes.data_as(_i32p)


def u32(a):
    return a.ctypes.data_as(_u32p)


def u16(a):
    return a.ctypes.data_as(_u16p)


def as_states(m, ndim_state):
    """C-contiguous uint8 array with a leading batch axis; returns (array, batched?)."""
    a = np.ascontiguousarray(m, dtype=np.uint8)
    if a.ndim == ndim_state:
        return a[None], False
    if a.ndim == ndim_state + 1:
        return a, True
    raise ValueError(f"expected a uint8 array of {ndim_state} or {ndim_state + 1} dimensions, got shape {a.shape}")


def make_params(code=TORIC, L=0, Nc=1, p=0.1, p_logical=0.0, iters=10, steps=0, tops_burn=2, TOPS=10, SEQ=2,
                eps=0.1, seed=0, first_syndrome=0, conv_mode=CONV_NONE, scan=SCAN_RANDOM,
                noise=NOISE_DEPOLARIZING, eta=0.0, alpha=0.0, device=0, replicas=0, flags=0):
    pr = Params()
    pr.abi_size = C.sizeof(Params)
    pr.code, pr.L, pr.Nc, pr.noise, pr.scan, pr.conv_mode, pr.device = code, L, Nc, noise, scan, conv_mode, device
    pr.iters, pr.steps, pr.tops_burn, pr.TOPS, pr.SEQ, pr.replicas = iters, steps, tops_burn, TOPS, SEQ, replicas
    pr.eps, pr.p, pr.eta, pr.alpha, pr.p_logical = eps, p, eta, alpha, p_logical
    pr.seed, pr.first_syndrome, pr.flags = seed & 0xFFFFFFFFFFFFFFFF, first_syndrome, int(flags)
    return pr
