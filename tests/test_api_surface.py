"""API-surface conformance of the drop-in mirrors ("decoders.py and generate_data.py drop in
unchanged").  tests/golden/f_api_surface.npz holds what the reference's callers -- decoders.py, decoders_biasednoise.py (run for a few
steps on recording proxies of the reference's own objects, tests/golden/gen_golden.py --only api) and generate_data.py (read as text) --
touch on the sampler path's classes: attribute names, call shapes, result kinds, constructor arities, and the names they take from
`src.*` by import.  Here: the qecmc modules a maintainer swaps in (`sys.modules["src.mcmc"] = qecmc.mcmc`, INTEGRATION.md Level 1) offer
all of it -- names and signatures without a GPU; result kinds and the PTEQ loop shape on the device (-m gpu)."""
import ast
import inspect
import os

import numpy as np
import pytest

import qecmc
import qecmc.decoders
import qecmc.decoders_biasednoise
import qecmc.mcmc
import qecmc.mcmc_alpha
import qecmc.mcmc_biased
import qecmc.planar_model
import qecmc.rotated_surface_model
import qecmc.toric_model
import qecmc.xzzx_model

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "f_api_surface.npz"))
MIRROR = {"src.mcmc": qecmc.mcmc, "src.mcmc_alpha": qecmc.mcmc_alpha, "src.mcmc_biased": qecmc.mcmc_biased, "src.toric_model": qecmc.toric_model,
          "src.xzzx_model": qecmc.xzzx_model, "src.rotated_surface_model": qecmc.rotated_surface_model, "src.planar_model": qecmc.planar_model}
CLASSES = {"Chain": qecmc.mcmc.Chain, "Ladder": qecmc.mcmc.Ladder, "Chain_xyz": qecmc.mcmc.Chain_xyz, "Chain_alpha": qecmc.mcmc_alpha.Chain_alpha,
           "Ladder_alpha": qecmc.mcmc_alpha.Ladder_alpha, "Chain_biased": qecmc.mcmc_biased.Chain_biased, "Ladder_biased": qecmc.mcmc_biased.Ladder_biased,
           "Toric_code": qecmc.Toric_code, "Planar_code": qecmc.Planar_code, "xzzx_code": qecmc.xzzx_code, "RotSurCode": qecmc.RotSurCode}


def _instances():
    """one small host-side instance per class (constructors touch no GPU)"""
    t, pl, x, r = qecmc.Toric_code(3), qecmc.Planar_code(3), qecmc.xzzx_code(3), qecmc.RotSurCode(3)
    return {"Toric_code": t, "Planar_code": pl, "xzzx_code": x, "RotSurCode": r,
            "Chain": qecmc.mcmc.Chain(0.1, t), "Ladder": qecmc.mcmc.Ladder(0.1, t, 3, 0.5), "Chain_xyz": qecmc.mcmc.Chain_xyz(np.array([0.05, 0.03, 0.04]), pl),
            "Chain_alpha": qecmc.mcmc_alpha.Chain_alpha(0.1, 2.0, x), "Ladder_alpha": qecmc.mcmc_alpha.Ladder_alpha(0.1, x, 2.0, 3, 0.5),
            "Chain_biased": qecmc.mcmc_biased.Chain_biased(0.1, x, 10.0) if _arity(qecmc.mcmc_biased.Chain_biased) >= 3 else None,
            "Ladder_biased": qecmc.mcmc_biased.Ladder_biased(0.1, x, 10.0, 3, 0.5)}


def _arity(cls):
    return len([p for p in inspect.signature(cls.__init__).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]) - 1


def _records():
    return [r.split("|") for r in FIX["records"]]


def test_every_name_the_reference_imports_from_src_exists_in_the_mirror_module():
    for rec in FIX["imports"]:
        importer, module, name = rec.split("|")
        assert hasattr(MIRROR[module], name), f"{importer} takes {name} from {module}: missing in {MIRROR[module].__name__}"
    # `from src.mcmc import *` (decoders.py:11) must deliver them without an __all__ that hides any
    ns = {}
    exec("from qecmc.mcmc import *", ns)
    assert {"Chain", "Ladder", "Chain_xyz"} <= set(ns)


def test_constructor_shapes():
    for rec in FIX["constructors"]:
        name, nargs, kwargs = rec.split("|")
        sig = inspect.signature(CLASSES[name].__init__)
        sig.bind(None, *([None] * int(nargs)), **{k: None for k in kwargs.split(",") if k})        # raises TypeError if the call shape does not fit


def test_every_attribute_and_call_shape_the_decoders_use():
    inst = _instances()
    for cls, attr, kind, nargs, kwargs, result in _records():
        obj = inst[cls]
        assert obj is not None and hasattr(obj, attr), f"{cls}.{attr} ({kind}) is used by the reference's decoders"
        if kind == "call":
            fn = getattr(obj, attr)
            assert callable(fn), f"{cls}.{attr} must be callable"
            inspect.signature(fn).bind(*([None] * int(nargs)), **{k: None for k in kwargs.split(",") if k})
        elif kind == "get" and result.startswith("ndarray"):
            v = getattr(obj, attr)
            nd, dt = result[len("ndarray"):].split(":")
            assert isinstance(v, np.ndarray) and v.ndim == int(nd) and str(v.dtype) == dt, f"{cls}.{attr}: {v!r} is not {result}"
        elif kind == "get" and result in ("int", "float", "list"):
            v = getattr(obj, attr)
            ok = {"int": (int, np.integer), "float": (float, np.floating, int), "list": (list,)}[result]
            assert isinstance(v, ok), f"{cls}.{attr} = {v!r} is not {result}"
        elif kind == "set":
            setattr(obj, attr, getattr(obj, attr))


def test_generate_data_code_attributes():
    for code in (qecmc.Toric_code(3), qecmc.Planar_code(3), qecmc.xzzx_code(3), qecmc.RotSurCode(3)):
        for attr in FIX["generate_data_code_attrs"]:
            assert hasattr(code, attr), f"generate_data.py uses init_code.{attr}"


def test_public_surface_of_the_reference_modules_is_mirrored():
    """Statically (the reference's files read as text in the build container only): every class and method of the sampler path's modules
    exists under the same name -- except the plotting helpers and the pandas data reader, which are outside the path."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("the reference is present in the build container only")
    skip_methods = {"plot"}
    skip_classes = {"MCMCDataReader"}
    files = {"src/mcmc.py": qecmc.mcmc, "src/mcmc_alpha.py": qecmc.mcmc_alpha, "src/mcmc_biased.py": qecmc.mcmc_biased, "src/toric_model.py": qecmc.toric_model,
             "src/xzzx_model.py": qecmc.xzzx_model, "src/rotated_surface_model.py": qecmc.rotated_surface_model, "src/planar_model.py": qecmc.planar_model}
    for f, mod in files.items():
        for node in ast.parse(open(os.path.join(ref, f)).read()).body:
            if isinstance(node, ast.ClassDef) and node.name not in skip_classes:
                cls = getattr(mod, node.name, None)
                assert cls is not None, f"{f}: class {node.name}"
                for m in node.body:
                    if isinstance(m, ast.FunctionDef) and m.name not in skip_methods:
                        assert hasattr(cls, m.name), f"{f}: {node.name}.{m.name}"
    # the decoders' entry points (their *_droplet helpers are internal to the reference's process pool: one fused launch here)
    for f, mod in (("decoders.py", qecmc.decoders), ("decoders_biasednoise.py", qecmc.decoders_biasednoise)):
        for node in ast.parse(open(os.path.join(ref, f)).read()).body:
            if isinstance(node, ast.FunctionDef) and not node.name.endswith("_droplet") and "droplet_" not in node.name:
                assert hasattr(mod, node.name), f"{f}: {node.name}"


# ---- on the device ----------------------------------------------------------------------------------------------------------------
def _kind(v):
    if isinstance(v, np.ndarray):
        return "ndarray%d:%s" % (v.ndim, v.dtype)
    if isinstance(v, tuple):
        return "tuple(" + ",".join(_kind(x) for x in v) + ")"
    if isinstance(v, (bool, np.bool_)):
        return "bool"
    if isinstance(v, (int, np.integer)):
        return "int"
    if isinstance(v, (float, np.floating)):
        return "float"
    return type(v).__name__


@pytest.mark.gpu
def test_result_kinds_of_every_recorded_call():
    """each call the decoders make, made on the mirrors with arguments of the recorded shape: the result is of the recorded kind"""
    inst = _instances()
    args = {("Toric_code", "to_class"): (3,), ("Chain", "update_chain_fast"): (5,), ("Chain_xyz", "update_chain_fast"): (5,), ("Chain_alpha", "update_chain"): (5,),
            ("Ladder", "step"): (10,), ("Ladder_alpha", "step"): (10,), ("Ladder_biased", "step"): (10,)}
    for cls, attr, kind, nargs, kwargs, result in _records():
        if kind != "call":
            continue
        a = args.get((cls, attr), ())
        assert len(a) == int(nargs), (cls, attr)
        got = getattr(inst[cls], attr)(*a)
        assert _kind(got) == result, f"{cls}.{attr}{a} returned {_kind(got)}, the reference's returns {result}"


@pytest.mark.gpu
@pytest.mark.parametrize("name,L,Nc", [("toric", 5, 5), ("xzzx", 5, 4)])
def test_reference_pteq_loop_shape_on_the_mirror_equals_the_batched_call(name, L, Nc):
    """decoders.PTEQ's loop as the reference writes it (decoders.py:55-68: ladder.step(iters), chains[0].code.define_equivalence_class(),
    .count_errors(), ladder.tops0) driven against qecmc.Ladder for 300 steps gives the class counts of ONE qecmc.pteq_batch launch
    of the same ladder, bit for bit: the drop-in classes and the batched path are the same chain."""
    rng = np.random.default_rng(5)
    code = qecmc.Toric_code(L) if name == "toric" else qecmc.xzzx_code(L)
    shape = code.qubit_matrix.shape
    code.qubit_matrix = (rng.integers(1, 4, size=shape) * (rng.random(shape) < 0.1)).astype(np.uint8)
    p, steps, iters, tops_burn, seed = 0.1, 300, 10, 1, 424242
    ladder = qecmc.Ladder(p, code, Nc, 0.5, seed=seed, stream=7)
    eq = np.zeros(code.nbr_eq_classes, np.uint32)
    series = []
    for step in range(steps):
        ladder.step(iters)
        current_eq = ladder.chains[0].code.define_equivalence_class()
        if ladder.tops0 >= tops_burn:
            eq[current_eq] += 1
            series.append(ladder.chains[0].code.count_errors())
    res = qecmc.pteq_batch(code.qubit_matrix[None], p, Nc=Nc, steps=steps, iters=iters, tops_burn=tops_burn, seed=seed, first_syndrome=7,
                           code=qecmc.TORIC if name == "toric" else qecmc.XZZX, return_states=True)
    assert np.array_equal(res["counts"][0], eq) and int(res["samples"][0]) == len(series) and int(res["tops0"][0]) == ladder.tops0
    assert np.array_equal(res["states"][0], np.stack([ch.code.qubit_matrix for ch in ladder.chains]))
