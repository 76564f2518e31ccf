"""Host symbolic analysis timing at C1 (usage: time_symbolic.py [threads...]); PLFEM_SYM_TRACE=1 adds sub-phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0)
m = generate_mesh(g, 1.0, 1)
for nth in [int(a) for a in sys.argv[1:]] or (1, 4, 8, 12, 16, 24):
    ts = []
    for rep in range(7):
        t0 = time.perf_counter(); s = _native.Symbolic(m.p, m.t, nthreads=nth); t1 = time.perf_counter(); ts.append(t1 - t0)
    i = s.info
    ts.sort()
    print(nth, 'min %.1f median %.1f ms' % (ts[0] * 1e3, ts[3] * 1e3), '| last: num %.1f pat %.1f tree %.1f fronts %.1f' % (i['t_numbering_us'] / 1e3, i['t_pattern_us'] / 1e3, i['t_tree_us'] / 1e3, i['t_fronts_us'] / 1e3), flush=True)
