"""MI355X-native vectorial H-field P2 FEM eigenmode path (drop-in for the reference ``solver_fem.py``)."""
from .geometry import MCFGeometry, PhotonicLanternGeometry, mcf_positions  # noqa: F401
from .mesh import TriMesh, generate_mesh  # noqa: F401

__all__ = ["MCFGeometry", "PhotonicLanternGeometry", "mcf_positions", "TriMesh", "generate_mesh"]
