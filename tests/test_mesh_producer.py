"""Mesh producer: native uniform refinement vs the oracle's restatement of
``MeshTri.refined()``, the ``MeshGenerator`` cache semantics of reference ``mesh.py:82-220``, and the
quality metrics of ``mesh.py:419-496``."""
import numpy as np
import pytest

from oracle.p2 import MeshTriLite, P2Basis
from pl_fem_vectoriel_amd import MCFGeometry, _native
from pl_fem_vectoriel_amd.mesh import (MeshGenerator, MeshQualityAnalyzer, SimulationConfig, TriMesh, generate_mesh,
                                       unit_square_mesh)


@pytest.mark.parametrize("which", ["square", "lantern"])
def test_native_refinement_is_bit_exact_with_the_oracle(built_library, c1_geometry, which):
    mesh = unit_square_mesh(5) if which == "square" else generate_mesh(c1_geometry, 0.4, 0)
    ref = MeshTriLite(mesh.p, mesh.t).refined(2)
    got = mesh.refined(2)
    np.testing.assert_array_equal(got.p, ref.p)
    np.testing.assert_array_equal(got.t, ref.t)
    assert got.nelements == 16 * mesh.nelements
    # area is conserved and the new vertices are the P2 nodes of the coarse mesh
    assert abs(P2Basis(ref).absdet.sum() - P2Basis(MeshTriLite(mesh.p, mesh.t)).absdet.sum()) < 1e-10
    p1, _ = _native.mesh_refine(mesh.p, mesh.t)
    np.testing.assert_array_equal(p1, P2Basis(MeshTriLite(mesh.p, mesh.t)).doflocs)
    with pytest.raises(ValueError):
        _native.mesh_refine(mesh.p, np.array([[0], [1], [10 ** 6]]))


def test_mesh_generator_cache_semantics(built_library):
    MeshGenerator.clear_cache()
    cfg = SimulationConfig(mesh_min_points=2000, mesh_target_points=6000, cache_max_size=2)
    geoms = [MCFGeometry(n, 8.0, 1.5, 1.535, 1.0) for n in (2, 3, 4)]
    m0, b0 = MeshGenerator.generate(geoms[0], 0.3, cfg)
    assert b0.N == m0.nvertices + m0.edges()[0].shape[1] and m0.nvertices >= 2000
    again = MeshGenerator.generate(geoms[0], 0.3, cfg)
    assert again[0] is m0                                              # cache hit returns the same objects
    MeshGenerator.generate(geoms[1], 0.3, cfg)
    MeshGenerator.generate(geoms[2], 0.3, cfg)                         # evicts the oldest entry (FIFO, size limit 2)
    st = MeshGenerator.get_cache_stats()
    assert st["size"] == 2 and st["hits"] == 1 and st["misses"] == 3 and 0 < st["hit_rate"] < 1
    assert MeshGenerator.generate(geoms[0], 0.3, cfg)[0] is not m0     # was evicted
    assert MeshGenerator.generate(geoms[0], 0.35, cfg)[0] is not m0    # refinement is part of the key
    off = SimulationConfig(enable_mesh_cache=False, mesh_min_points=100, mesh_target_points=300)
    a = MeshGenerator.generate(geoms[1], 0.3, off)[0]
    b = MeshGenerator.generate(geoms[1], 0.3, off)[0]
    assert a is not b
    MeshGenerator.clear_cache()
    assert MeshGenerator.get_cache_stats()["size"] == 0


def test_default_config_reproduces_the_north_star_mesh(built_library, c1_geometry):
    MeshGenerator.clear_cache()
    mesh, basis = MeshGenerator.generate(c1_geometry, 1.0)
    assert (mesh.nvertices, mesh.nelements, basis.N) == (22694, 45252, 90639)          # config C1
    assert len(basis.get_dofs().all()) == 268
    MeshGenerator.clear_cache()


def test_quality_metrics():
    m = TriMesh(np.array([[0, 1, 0.5, 2.0], [0, 0, np.sqrt(3) / 2, 0.0]]), np.array([[0, 1], [1, 3], [2, 2]]))
    q = MeshQualityAnalyzer.analyze(m)
    assert q["n_points"] == 4 and q["n_elements"] == 2
    assert abs(q["quality_max"] - 1.0) < 1e-9 and abs(q["aspect_min"] - 1.0) < 1e-9     # the equilateral one
    assert abs(q["min_angle_mean"] - (60 + q["min_angle_min"]) / 2) < 1e-9
    assert q["area_min"] > 0 and 0 <= q["poor_quality_frac"] <= 1
    assert MeshQualityAnalyzer.analyze(None) == {}


def test_cache_file_round_trip_and_stats_table(built_library, tmp_path, capsys, caplog):
    """``save_cache`` / ``load_cache`` / ``print_cache_stats`` of reference ``mesh.py:371-416``: the meshes and counters
    survive a round trip through a file (the P2 basis view is rebuilt), a missing file is a warning."""
    MeshGenerator.clear_cache()
    cfg = SimulationConfig(mesh_min_points=500, mesh_target_points=1500)
    g = MCFGeometry(3, 8.0, 1.5, 1.535, 1.0)
    m0, b0 = MeshGenerator.generate(g, 0.3, cfg)
    MeshGenerator.generate(g, 0.3, cfg)
    path = tmp_path / "meshes.pkl"
    MeshGenerator.save_cache(path)
    MeshGenerator.clear_cache()
    with caplog.at_level("WARNING"):
        MeshGenerator.load_cache(tmp_path / "absent.pkl")
    assert MeshGenerator.get_cache_stats()["size"] == 0 and "no mesh cache file" in caplog.text
    MeshGenerator.load_cache(path)
    st = MeshGenerator.get_cache_stats()
    assert (st["size"], st["hits"], st["misses"]) == (1, 1, 1)
    m1, b1 = MeshGenerator.generate(g, 0.3, cfg)                       # a hit on the restored entry
    np.testing.assert_array_equal(m1.p, m0.p)
    np.testing.assert_array_equal(m1.t, m0.t)
    assert b1.N == b0.N and MeshGenerator.get_cache_stats()["hits"] == 2
    MeshGenerator.print_cache_stats()
    out = capsys.readouterr().out
    assert "MESH CACHE" in out and "hit rate" in out and "66.7%" in out
    MeshGenerator.clear_cache()


def test_quality_validation_and_report(caplog):
    """``validate_mesh_quality`` / ``print_analysis`` of reference ``mesh.py:499-568``: thresholds 10 deg / 20 / 20 %, and
    20 deg / 3 / 0.7 in strict mode."""
    good = unit_square_mesh(4)                                         # right isosceles triangles: 45 deg, aspect 1.41
    ok, msg = MeshQualityAnalyzer.validate_mesh_quality(good)
    assert ok and "acceptable" in msg
    ok, msg = MeshQualityAnalyzer.validate_mesh_quality(good, strict=True)
    assert ok
    sliver = TriMesh(np.array([[0.0, 1.0, 0.02], [0.0, 0.0, 0.01]]), np.array([[0], [1], [2]]))   # a needle: 0.6 deg, aspect 45
    ok, msg = MeshQualityAnalyzer.validate_mesh_quality(sliver)
    assert not ok and "smallest angle" in msg and "aspect ratio" in msg and "poor-quality" in msg
    mid = TriMesh(np.array([[0.0, 1.0, 0.5], [0.0, 0.0, 0.17]]), np.array([[0], [1], [2]]))   # 18.8 deg, quality 0.38
    assert MeshQualityAnalyzer.validate_mesh_quality(mid)[0]
    ok_s, msg_s = MeshQualityAnalyzer.validate_mesh_quality(mid, strict=True)
    assert not ok_s and "[strict]" in msg_s
    assert MeshQualityAnalyzer.validate_mesh_quality(None) == (False, "invalid mesh (analysis failed)")
    with caplog.at_level("INFO"):
        MeshQualityAnalyzer.print_analysis(good)
    assert "MESH QUALITY" in caplog.text and "triangles" in caplog.text
    with caplog.at_level("WARNING"):
        MeshQualityAnalyzer.print_analysis(None)
    assert "nothing to analyse" in caplog.text
