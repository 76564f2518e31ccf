"""A/B of the analysis: side chain concurrent with the tree vs in sequence (PLFEM_SYM_SEQUENTIAL), interleaved repeats."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from pl_fem_vectoriel_amd import _native, MCFGeometry, generate_mesh
    if os.environ.get("MALLOPT"):
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20); libc.mallopt(-1, 1 << 30)
    geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(geom, 1.0, 1)
    for nt in (8, 16):
        ts = []
        for rep in range(40):
            t0 = time.perf_counter()
            s = _native.Symbolic(mesh.p, mesh.t, nthreads=nt)
            ts.append(1e3 * (time.perf_counter() - t0))
        ts = np.array(ts[5:])
        print(f"  nthreads {nt:2d}: min {ts.min():.2f} median {np.median(ts):.2f} p90 {np.percentile(ts, 90):.2f} ms   {s.info['t_numbering_us']} {s.info['t_pattern_us']} {s.info['t_tree_us']} {s.info['t_fronts_us']}")
else:
    for rnd in range(2):
        for seq in (0, 1):
            for mo in (0, 1):
                env = dict(os.environ)
                if seq: env["PLFEM_SYM_SEQUENTIAL"] = "1"
                if mo: env["MALLOPT"] = "1"
                print(f"sequential {seq} mallopt {mo}", flush=True)
                subprocess.run([sys.executable, __file__, "child"], env=env)
