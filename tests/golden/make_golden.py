#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REAL reference compiled in this container.

Run only where /root/reference exists:   make -C oracle ref && python tests/golden/make_golden.py

What it writes (data only -- inputs and expected outputs, never reference source):
  ref_ops.npz      seeded inputs + outputs of single reference operators
                   (Jacobi, Gauss-Seidel, Residual, interpolate, Solver::Solve, one
                   SawtoothMGIteration) obtained through oracle/_ref/ref_ops
  ref_solve.json   residual histories (17 s.d.), per-cycle coarse residuals (6 s.d.,
                   parsed from the reference's own stdout) and per-cycle coarse sweep counts
                   (counting subclasses of the reference smoothers) for whole solves
  ref_solve_u.npz  final solution vectors of the small whole solves and of BASELINE config 1 (n = 257)
  fixture_*.txt/.npz   the reference's own result files, copied as data:
                   GeometricMultigrid/test/{MGGS4.txt,x.mtx} (-n 385 -a 1 -w 10 -ml 5 -test 0 -smt 2)
                   WebInterface/{MGGS4.txt,x.mtx}            (-n 145 -a 1 -w 10 -ml 5 -test 1 -smt 1)
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF_OPS = os.path.join(ROOT, "oracle", "_ref", "ref_ops")
REFDIR = "/root/reference"


def run_op(op, n, levels, level, alpha, length, smt, test, u=None, b=None):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        if u is not None:
            np.concatenate([u.ravel(), b.ravel()]).astype("<f8").tofile(fin)
        else:
            open(fin, "wb").close()
        subprocess.run([REF_OPS, op, str(n), str(levels), str(level), repr(alpha), repr(length),
                        str(smt), str(test), fin, fout], check=True)
        return np.fromfile(fout, "<f8")


def main():
    if not os.path.exists(REF_OPS):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    ops = {}
    meta = []
    rng = np.random.default_rng(20260101)
    for (n, levels, alpha, length) in [(17, 3, 1.0, 10.0), (33, 4, 2.5, 4.0), (25, 2, 0.7, 1.0)]:
        key = f"n{n}_L{levels}"
        u = rng.standard_normal((n, n))
        b = rng.standard_normal((n, n))
        ops[f"{key}_u"] = u
        ops[f"{key}_b"] = b
        meta.append(dict(key=key, n=n, levels=levels, alpha=alpha, length=length))
        for l in range(levels):
            st = 2 ** l
            o = run_op("jacobi", n, levels, l, alpha, length, 1, 0, u, b).reshape(n, n)
            ops[f"{key}_jacobi_l{l}"] = o[::st, ::st].copy()
            o = run_op("gs", n, levels, l, alpha, length, 0, 0, u, b).reshape(n, n)
            ops[f"{key}_gs_l{l}"] = o[::st, ::st].copy()
            o = run_op("residual", n, levels, l, alpha, length, 0, 0, u, b)
            ops[f"{key}_residual_l{l}"] = o[:n * n].reshape(n, n)[::st, ::st].copy()
            ops[f"{key}_residual_norm_l{l}"] = o[n * n]
            if l < levels - 1:
                o = run_op("interp", n, levels, l, alpha, length, 0, 0, u, b).reshape(n, n)
                ops[f"{key}_interp_l{l}"] = o[::st, ::st].copy()
        lc = levels - 1
        st = 2 ** lc
        for smt in (0, 1):
            o = run_op("coarse_solve", n, levels, lc, alpha, length, smt, 0, np.zeros_like(u), b)
            ops[f"{key}_coarse_smt{smt}_e"] = o[:n * n].reshape(n, n)[::st, ::st].copy()
            ops[f"{key}_coarse_smt{smt}_stats"] = o[n * n:n * n + 3]  # Norm, sweeps, Status
            o = run_op("cycle", n, levels, 0, alpha, length, smt, 0, u, b).reshape(n, n)
            ops[f"{key}_cycle_smt{smt}"] = o
    ops["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_ops.npz"), **ops)

    solves = []
    us = {}
    cases = [  # n alpha w ml test smt   (SURVEY §8c table + the two fixture runs)
        (17, 1, 10, 2, 0, 0), (17, 1, 10, 2, 0, 1), (17, 1, 10, 3, 1, 1), (33, 1, 10, 3, 1, 0),
        (33, 1, 10, 3, 2, 1), (65, 2.5, 4, 4, 1, 1), (257, 1, 10, 3, 1, 1), (257, 1, 10, 3, 1, 0),
        (145, 1, 10, 5, 1, 1), (385, 1, 10, 5, 0, 2), (65, 1, 10, 1, 0, 1), (129, 3, 2, 6, 2, 0),
    ]
    for (n, a, w, ml, test, smt) in cases:
        o = run_op("solve_full", n, ml, 0, float(a), float(w), smt, test)
        nh = int(o[0])
        hist = o[1:1 + nh]
        coarse = o[1 + nh:1 + nh + (nh - 1)]
        u = o[1 + nh + (nh - 1):1 + nh + (nh - 1) + n * n]
        key = f"n{n}_a{a}_w{w}_ml{ml}_t{test}_s{smt}"
        # the same run with counting smoothers (ref_harness.cpp: Counted<>): sweeps the coarse Solver
        # spent in every cycle -- what mg_solve_lockstep replays -- and the same history again
        oc = run_op("solve_counts", n, ml, 0, float(a), float(w), smt, test)
        assert int(oc[0]) == nh and np.array_equal(oc[1:1 + nh], hist), key
        counts = oc[1 + nh:1 + nh + (nh - 1)]
        assert np.array_equal(oc[1 + nh + (nh - 1):1 + nh + (nh - 1) + n * n], u), key
        solves.append(dict(key=key, n=n, alpha=a, length=w, levels=ml, test=test, smt=smt,
                           hist=[repr(float(x)) for x in hist],
                           coarse_relres=[repr(float(x)) for x in coarse],
                           coarse_counts=[int(c) for c in counts]))
        if n <= 65 or n == 257:   # 257: BASELINE config 1, the final solution vector north_star asks for
            us[key] = u.reshape(n, n)
        print(key, nh, hist[-1])
    with open(os.path.join(HERE, "ref_solve.json"), "w") as f:
        json.dump(solves, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "ref_solve_u.npz"), **us)

    # the reference's own result files, as data
    shutil.copyfile(f"{REFDIR}/GeometricMultigrid/test/MGGS4.txt", os.path.join(HERE, "fixture_gmgtest_MGGS4.txt"))
    shutil.copyfile(f"{REFDIR}/WebInterface/MGGS4.txt", os.path.join(HERE, "fixture_web_MGGS4.txt"))
    for src, dst in ((f"{REFDIR}/GeometricMultigrid/test/x.mtx", "fixture_gmgtest_x.npz"),
                     (f"{REFDIR}/WebInterface/x.mtx", "fixture_web_x.npz")):
        vals = np.loadtxt(src)
        cnt = int(vals[0])
        assert cnt == vals.size - 1
        np.savez_compressed(os.path.join(HERE, dst), x=vals[1:])
    for p in os.listdir(HERE):
        os.chmod(os.path.join(HERE, p), 0o644)


if __name__ == "__main__":
    main()
