"""Analysis time against the worker count (box: 16 cores per GPU), median of 30, with the phase split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pl_fem_vectoriel_amd import _native, MCFGeometry, generate_mesh
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for nt in (1, 4, 8, 12, 16, 24, 32):
    ts, infos = [], []
    for rep in range(30):
        t0 = time.perf_counter()
        s = _native.Symbolic(mesh.p, mesh.t, nthreads=nt)
        ts.append(time.perf_counter() - t0)
        infos.append(s.info)
    med = lambda k: int(np.median([i[k] for i in infos[3:]]))
    print(f"nthreads {nt:2d}: min {1e3 * min(ts):6.2f} median {1e3 * np.median(ts[3:]):6.2f} ms  numbering {med('t_numbering_us')} pattern {med('t_pattern_us')} tree {med('t_tree_us')} fronts {med('t_fronts_us')}", flush=True)
