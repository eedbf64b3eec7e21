"""Host-side logic of the Python mirror that needs no GPU."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

import qecmc


def test_percent_vector_reproduces_reference_truncation():
    # decoders.py:89 divides in float64, multiplies by 100 and truncates: 29/100*100 -> 28.999.. -> 28
    counts = np.zeros(16, dtype=np.uint32)
    counts[3], counts[7] = 29, 71
    pct = qecmc.percent_from_counts(counts, 100)
    assert pct[3] == np.uint8(np.divide(29, 100) * 100) == 28 and pct[7] == 71
    # nothing recorded (burn-in never ended): all-zero vector, as eq[0]/1
    assert not qecmc.percent_from_counts(np.zeros(16, dtype=np.uint32), 0).any()
    batch = qecmc.percent_from_counts(np.array([[1, 2, 0], [0, 0, 3]], dtype=np.uint32), np.array([3, 3]))
    assert batch.tolist() == [[33, 66, 0], [0, 0, 100]]


def test_ladder_probabilities_match_the_reference():
    g = np.load(os.path.join(GOLDEN, "f2_toric.npz"))
    for case in [c for c in g["cases"] if str(c).startswith("ladder")]:
        L, p, Nc, iters, nstep, seed, ndraw = g[f"{case}_par"]
        code = qecmc.Toric_code(int(L))
        code.qubit_matrix = g[f"{case}_init"].copy()
        ld = qecmc.Ladder(float(p), code, int(Nc), 0.5, seed=1)
        assert np.array_equal(ld.p_ladder, g[f"{case}_p_ladder"]) and np.array_equal(ld.p_diff, g[f"{case}_p_diff"])
        assert [c.flag for c in ld.chains] == [0] * (int(Nc) - 1) + [1] and ld.chains[-1].p_logical == 0.5
        assert ld.chains[0].factor == (float(p) / 3.0) / (1.0 - float(p))
        assert all(np.array_equal(c.code.qubit_matrix, code.qubit_matrix) for c in ld.chains)


def test_objects_are_plain_numpy_and_picklable():
    import copy
    import pickle
    code = qecmc.Toric_code(5)
    code.qubit_matrix[0, 1, 2] = 3
    ld = qecmc.Ladder(0.1, code, 5, 0.5, seed=3)
    ld2 = pickle.loads(pickle.dumps(copy.deepcopy(ld)))
    assert ld2.seed == 3 and np.array_equal(ld2.chains[2].code.qubit_matrix, code.qubit_matrix)


def test_generate_random_error_follows_numpy_global_rng():
    # same draw order as toric_model.py:15-23 (uniform then randint, per layer); syndrom() needs the GPU,
    # so only the matrix is checked here
    np.random.seed(1)
    exp = np.zeros((2, 5, 5), dtype=np.uint8)
    for i in range(2):
        q = np.random.uniform(0, 1, size=(5, 5))
        pauli = np.random.randint(3, size=(5, 5)) + 1
        exp[i] = np.where(q < 0.1, pauli, 0)
    g = np.load(os.path.join(GOLDEN, "f4_config1.npz"))
    assert np.array_equal(exp, g["init"])            # the reference's own matrix under np.random.seed(1)


def test_batch_metrics_line_is_json_and_consistent():
    """The per-batch metrics line from a synthetic result: proposals, rates, swap acceptance, tops0 histogram."""
    import json
    from qecmc import harness, _lib as L_
    res = dict(counts=np.zeros((4, 16), dtype=np.uint32), samples=np.array([10, 0, 5, 7], dtype=np.uint32),
               tops0=np.array([0, 3, 25, 1], dtype=np.uint32), steps_done=np.array([100, 100, 50, 100], dtype=np.uint32),
               converged=np.array([False, False, True, False]), stats=dict(kernel_ms=2.0),
               swap_accepts=np.array([[10, 5], [20, 5], [5, 0], [15, 10]], dtype=np.uint32),
               nerr_sums=np.array([[100, 200, 300]] * 4, dtype=np.uint32))
    m = harness.batch_metrics(L_.TORIC, 5, 3, 10, res, wall_s=0.01, success=np.array([True, False, True, True]))
    m = json.loads(json.dumps(m))
    assert m["proposals"] == 350 * 3 * 10 and m["syndromes"] == 4
    assert m["chain_sweeps_per_s_kernel"] == pytest.approx(350 * 30 / 50 / 2e-3)
    assert m["tops0_hist"][0] == 1 and m["tops0_hist"][20] == 1 and sum(m["tops0_hist"]) == 4
    assert m["swap_acceptance"] == pytest.approx([50 / 350, 20 / 350]) and m["mean_errors_per_rung"][2] == pytest.approx(1200 / 350)
    assert m["success_rate"] == 0.75 and m["converged_frac"] == 0.25 and m["frac_past_burn_in"] == 0.75


def test_shard_names_and_bench_configs():
    from qecmc import harness
    assert harness.shard_name("data", 3, 12) == "data_seed3_shard00012.npz"
    import bench
    for c, (code, L, N) in {2: ("toric", 9, 65536), 3: ("toric", 15, 131072), 4: ("xzzx", 9, 65536), 5: ("rotated", 21, 32768)}.items():
        a = bench.parse_args(["--config", str(c)])
        assert (a.code, a.L, a.syndromes, a.Nc, a.ladder_steps) == (code, L, N, 8, 10000)
    assert bench.parse_args(["--config", "4"]).eta == 100.0 and bench.parse_args([]).config == 2
    assert bench.parse_args(["--config", "3", "--Nc", "15"]).Nc == 15
