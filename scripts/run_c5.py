import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
g = MCFGeometry(19, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
t0=time.time(); mesh = generate_mesh(g, 1.0, 2); print('mesh', mesh.nvertices, mesh.nelements, 'gen %.1fs'%(time.time()-t0), flush=True)
s = TrueVectorialMaxwellSolver(g, device=0)
for rep in range(2):
    s.clear_cache(); t0=time.time(); modes = s.solve_vectorial_modes(mesh, 20); t1=time.time()
    st = s.last_stats
    print('C5 solve %.1f ms'%((t1-t0)*1e3), len(modes), {k:(round(v,3) if isinstance(v,float) else v) for k,v in st.items() if k in ('N','n','n_req','ncv','nconv','n_opinv','n_block_solves','restarts','t_symbolic','t_context','factor_us','lanczos_us','pivot_perturbations')}, flush=True)
print('mem GB', torch.cuda.max_memory_allocated()/1e9)
print([round(m['n_eff'],6) for m in modes[:6]])
