"""The checker's field comparison: sign- and degenerate-basis-invariant, but not blind."""
import numpy as np

from oracle.compare import column_errors, mode_field_errors


def test_isolated_columns_sign_invariant_and_sensitive():
    rng = np.random.default_rng(0)
    U = rng.standard_normal((50, 3))
    keys = np.array([3.0, 2.0, 1.0])
    V = U * np.array([1.0, -1.0, 2.5])                               # sign and scale do not matter
    assert column_errors(V, U, keys, 1e-6).max() < 1e-14
    V[:, 1] += 1e-3 * rng.standard_normal(50)
    e = column_errors(V, U, keys, 1e-6)
    assert e[0] < 1e-14 and e[2] < 1e-14 and 1e-4 < e[1] < 1e-2


def test_cluster_is_compared_as_a_subspace():
    rng = np.random.default_rng(1)
    U = np.linalg.qr(rng.standard_normal((40, 3)))[0]
    keys = np.array([2.0, 1.0 + 1e-9, 1.0])
    c, s = np.cos(0.7), np.sin(0.7)
    V = U.copy()
    V[:, 1], V[:, 2] = c * U[:, 1] + s * U[:, 2], -s * U[:, 1] + c * U[:, 2]
    assert column_errors(V, U, keys, 1e-6).max() < 1e-14              # rotated inside the pair: same space
    assert column_errors(V, U, keys, 1e-12)[1] > 0.1                  # treated as isolated: differs
    V[:, 2] = U[:, 0]                                                 # leaves the pair's space
    assert column_errors(V, U, keys, 1e-6)[1] > 0.5


def test_mode_records_wrapper():
    rng = np.random.default_rng(2)
    ref = [{"n_eff": 1.5 - 0.01 * i, "Ex_dofs": rng.standard_normal(9), "Ey_dofs": rng.standard_normal(9)} for i in range(4)]
    got = [{"n_eff": m["n_eff"], "Ex_dofs": -m["Ex_dofs"], "Ey_dofs": -m["Ey_dofs"]} for m in ref]
    assert mode_field_errors(got, ref).max() < 1e-14
    assert mode_field_errors([], []).size == 0
