"""Host eigensolver of the Lanczos drivers (csrc/host_eig.cpp) against numpy.linalg.eigh: full decomposition
and the last-rows-only form the per-step convergence test uses.  CPU only."""
import numpy as np
import pytest

from pl_fem_vectoriel_amd import _native


def _band_clustered(n, rng, hbw=4):
    """Shape of a block-Lanczos projected matrix: banded, with pairs of eigenvalues 1e-9 apart."""
    a = np.zeros((n, n))
    d = np.repeat(rng.standard_normal((n + 1) // 2), 2)[:n]
    a[np.arange(n), np.arange(n)] = d * (1 + 1e-9 * rng.standard_normal(n))
    for k in range(1, hbw + 1):
        v = 0.3 * rng.standard_normal(n - k)
        a[np.arange(n - k), np.arange(k, n)] = v
        a[np.arange(k, n), np.arange(n - k)] = v
    return a


@pytest.mark.parametrize("n", [1, 2, 3, 7, 44, 104, 164])
@pytest.mark.parametrize("kind", ["dense", "band"])
def test_full_decomposition(built_library, n, kind):
    rng = np.random.default_rng(100 + n)
    m = rng.standard_normal((n, n))
    a = (m + m.T) / 2 if kind == "dense" or n < 7 else _band_clustered(n, rng)
    w, v = _native.debug_symeig(a)
    s = v.T                                                            # columns = eigenvectors
    scale = max(1.0, np.abs(a).max())
    assert np.abs(np.sort(w) - np.linalg.eigvalsh(a)).max() < 5e-14 * scale * n
    assert np.abs(a @ s - s * w).max() < 5e-14 * scale * n
    assert np.abs(s.T @ s - np.eye(n)).max() < 5e-14 * n


@pytest.mark.parametrize("n,p", [(1, 1), (4, 4), (5, 4), (44, 4), (104, 4), (164, 4), (60, 1), (60, 8)])
def test_last_rows_match_the_full_vectors(built_library, n, p):
    rng = np.random.default_rng(7 * n + p)
    a = _band_clustered(n, rng) if n >= 44 else (lambda m: (m + m.T) / 2)(rng.standard_normal((n, n)))
    w, v = _native.debug_symeig(a)
    w2, y = _native.debug_symeig(a, last_rows=p)
    assert y.shape == (n, p)
    np.testing.assert_array_equal(w, w2)                                # same reduction, same QL sweeps
    # same rotations applied to the same starting rows: identical up to rounding, signs included
    assert np.abs(v[:, n - p:] - y).max() < 1e-13
    # and the quantity the driver uses, || R y_i || for an upper-triangular R, agrees to rounding
    r = np.triu(rng.standard_normal((p, p)))
    assert np.abs(np.linalg.norm(v[:, n - p:] @ r.T, axis=1) - np.linalg.norm(y @ r.T, axis=1)).max() < 1e-13


def test_rejects_bad_arguments(built_library):
    with pytest.raises(ValueError):
        _native.debug_symeig(np.zeros((3, 4)))
    with pytest.raises(ValueError):
        _native.debug_symeig(np.eye(3), last_rows=5)


@pytest.mark.parametrize("n,nsel", [(1, 1), (3, 3), (8, 4), (44, 22), (100, 22), (104, 32), (164, 22)])
def test_band_path_selected_vectors(built_library, n, nsel):
    """Band reduction + QL eigenvalues + inverse iteration for the eigenvalues of largest magnitude (the final
    Ritz decomposition of a run without restart): eigenvalues of the band part, residuals and orthonormality of
    the selected vectors — including pairs 1e-9 apart — at the level of the dense solver."""
    rng = np.random.default_rng(31 * n + nsel)
    b = 4
    a = _band_clustered(n, rng, hbw=b) if n >= 8 else np.diag(rng.standard_normal(n))
    noise = 1e-17 * rng.standard_normal((n, n))            # what full reorthogonalisation leaves outside the band
    w, v = _native.debug_symeig_band(a + np.triu(noise, b + 1) + np.triu(noise, b + 1).T, b, nsel)
    scale = max(1.0, np.abs(a).max())
    assert (np.diff(w) >= 0).all()
    assert np.abs(w - np.linalg.eigvalsh(a)).max() < 5e-14 * scale * n
    sel = np.argsort(-np.abs(w), kind="stable")[:nsel]
    s = v[sel]
    assert np.abs(np.delete(v, sel, axis=0)).max(initial=0.0) == 0.0
    assert np.abs(s @ a - s * w[sel][:, None]).max() < 5e-14 * scale * n
    assert np.abs(s @ s.T - np.eye(nsel)).max() < 5e-14 * n


def test_band_path_exactly_repeated_eigenvalues(built_library):
    """Two decoupled copies of the same band matrix: every eigenvalue is exactly double."""
    rng = np.random.default_rng(5)
    blk = _band_clustered(24, rng)
    a = np.zeros((48, 48))
    a[:24, :24] = blk
    a[24:, 24:] = blk
    w, v = _native.debug_symeig_band(a, 4, 20)
    sel = np.argsort(-np.abs(w), kind="stable")[:20]
    s = v[sel]
    assert np.abs(s @ a - s * w[sel][:, None]).max() < 1e-12
    assert np.abs(s @ s.T - np.eye(20)).max() < 1e-12
