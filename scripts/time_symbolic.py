import numpy as np, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0)
m = generate_mesh(g, 1.0, 1)
for nth in (1,4,8,12,16,24):
    best=1e9
    for rep in range(3):
        t0=time.perf_counter(); s=_native.Symbolic(m.p, m.t, nthreads=nth); t1=time.perf_counter(); best=min(best,t1-t0)
    i=s.info
    print(nth, 'total %.1f ms'%(best*1e3), 'num %.1f pat %.1f tree %.1f fronts %.1f'%(i['t_numbering_us']/1e3,i['t_pattern_us']/1e3,i['t_tree_us']/1e3,i['t_fronts_us']/1e3))
