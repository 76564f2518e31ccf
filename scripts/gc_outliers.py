"""Are the slow steps of the cold-solve loop Python's cyclic garbage collector?  60 cold C1 solves with a gc callback that
records every collection (generation, duration, objects collected); prints the steps beside the collections that fell into
them, then the same loop with the collector disabled."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PLFEM_MALLOC_TUNE", "1")
import torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver

geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, 1)
events = []
state = {}
def cb(phase, info):
    if phase == "start":
        state["t"] = time.perf_counter()
    else:
        events.append((time.perf_counter(), info["generation"], (time.perf_counter() - state["t"]) * 1e3, info["collected"]))
gc.callbacks.append(cb)

def loop(n):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        s = TrueVectorialMaxwellSolver(geom, device=0, reuse_symbolic=False)
        s.solve_vectorial_modes(mesh, 10)
        t1 = time.perf_counter()
        out.append((t0, t1, s.last_stats["t_symbolic"] * 1e3))
    return out

def cgroup():
    out = {}
    for name in ("cpu.max", "cpu.stat"):
        for root in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
            try:
                out[name] = open(os.path.join(root, name)).read().split()
                break
            except OSError:
                pass
    return out

print("cgroup at start:", cgroup(), "cpus allowed:", len(os.sched_getaffinity(0)))
loop(5)
for label, prep in (("collector on", lambda: gc.enable()), ("collector off", lambda: (gc.collect(), gc.disable()))):
    prep()
    events.clear()
    steps = loop(60)
    ms = [(b - a) * 1e3 for a, b, _ in steps]
    print(f"== {label}: median {sorted(ms)[30]:.2f} ms, mean {sum(ms) / 60:.2f}, max {max(ms):.2f}; collections: "
          f"{[sum(1 for e in events if e[1] == g) for g in (0, 1, 2)]} (gen 0 / 1 / 2)")
    for i, (a, b, sym) in enumerate(steps):
        ev = [e for e in events if a <= e[0] <= b]
        if ms[i] > sorted(ms)[30] + 0.8 or any(e[2] > 0.2 for e in ev):
            print(f"  step {i:2d}: {ms[i]:6.2f} ms  analysis {sym:5.2f} ms  gc: " + ", ".join(f"gen{e[1]} {e[2]:.2f} ms ({e[3]} objects)" for e in ev))
gc.enable()
print("cgroup at end:", cgroup())
