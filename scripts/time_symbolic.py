"""Wall time of the mesh-only analysis (plfem_symbolic_create) at C1, with the sub-phase trace of the last repeat."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pl_fem_vectoriel_amd import _native, MCFGeometry, generate_mesh

if os.environ.get("MALLOPT"):          # experiment: keep freed blocks in the heap (no mmap / munmap per analysis)
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    print("mallopt", libc.mallopt(-3, 32 << 20),          # M_MMAP_THRESHOLD
          libc.mallopt(-1, 1 << 30))     # M_TRIM_THRESHOLD
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, levels)
ts = []
for rep in range(12):
    if rep == 11:
        os.environ["PLFEM_SYM_TRACE"] = "1"
    t0 = time.perf_counter()
    s = _native.Symbolic(mesh.p, mesh.t)
    ts.append(1e3 * (time.perf_counter() - t0))
print("symbolic ms:", " ".join(f"{t:.2f}" for t in ts))
print({k: v for k, v in s.info.items() if k.startswith("t_")})
