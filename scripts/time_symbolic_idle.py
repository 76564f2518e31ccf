"""Does the analysis slow down when the host was idle before it (as it is while the GPU runs the previous solve)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pl_fem_vectoriel_amd import _native, MCFGeometry, generate_mesh
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, 1)
for nt in (8, 16):
    for idle_ms in (0, 5, 20, 20, 0):
        ts = []
        for rep in range(25):
            if idle_ms:
                time.sleep(idle_ms * 1e-3)
            t0 = time.perf_counter()
            s = _native.Symbolic(mesh.p, mesh.t, nthreads=nt)
            ts.append(time.perf_counter() - t0)
        print(f"nthreads {nt:2d} idle {idle_ms:2d} ms before each build: median {1e3 * np.median(ts[3:]):.2f} ms  min {1e3 * min(ts):.2f}", flush=True)
