"""Times mg_set_rhs / mg_get_solution at 513^3 fp64 (the PCIe-inclusive figure DESIGN.md quotes)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from multigrid_prj_amd import capi
n = 513
with capi.Solver(capi.make_desc(dim=3, n=n, levels=6, cycle=capi.CYCLE_V, smoother=capi.SMOOTH_JACOBI, omega=6/7, nu_pre=2, nu_post=2,
                                restriction=capi.RESTRICT_FULLW, outer_pre_gs=0)) as s:
    b = np.random.default_rng(0).random((n, n, n))
    for k in range(2):
        t0 = time.perf_counter(); s.set_rhs(b); t1 = time.perf_counter(); u = s.get_solution(); t2 = time.perf_counter()
        print(f"set_rhs {t1 - t0:.3f} s ({b.nbytes / (t1 - t0) / 1e9:.1f} GB/s), get_solution {t2 - t1:.3f} s ({b.nbytes / (t2 - t1) / 1e9:.1f} GB/s)")
