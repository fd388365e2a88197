"""BASELINE config 1 (257^2, 3 levels, Jacobi) through the C-ABI: cold and warm solve times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from multigrid_prj_amd import capi

n = 257
m_h = 10.0 / (n - 1)
j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
x, y = i * m_h, 10.0 - j * m_h
bnd = (i == 0) | (j == 0) | (i == n - 1) | (j == n - 1)
b = np.where(bnd, np.exp(x) * np.exp(-2 * y), -5.0 * np.exp(x) * np.exp(-2 * y))
with capi.Solver(capi.make_desc(dim=2, n=n, levels=3, alpha=1.0, length=10.0, smoother=capi.SMOOTH_JACOBI)) as s:
    for k in range(3):
        s.set_rhs(b); s.zero_array(capi.ARR_U, 0); s.sync()
        t0 = time.perf_counter()
        hist, st = s.solve(1e-11, 1000)
        t1 = time.perf_counter()
        print(f"solve {k}: {1e3 * (t1 - t0):.2f} ms, {len(hist) - 1} cycles, coarse iterations {sum(x.coarse_iters for x in st)}")
