"""Host-side logic of the Python mirror that needs no GPU."""
import os

import numpy as np

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
