"""Pins the CPU oracle (oracle/gmg_oracle.c) to the REAL reference.

Golden data: tests/golden/ref_ops.npz, ref_solve.json, ref_solve_u.npz were produced
by the unmodified reference classes compiled in the build container
(oracle/ref_harness.cpp + tests/golden/make_golden.py); fixture_* are the reference's
own committed result files. CPU only, no GPU, no /root/reference at run time.
"""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ref_ops():
    z = np.load(os.path.join(G, "ref_ops.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def _desc(m, smoother=po.SMOOTH_JACOBI):
    return po.make_desc(dim=2, n=m["n"], levels=m["levels"], alpha=m["alpha"], length=m["length"],
                        smoother=smoother)


def test_single_operators_bit_exact(ref_ops):
    """Jacobi / GS / Residual / interpolate of the reference, level by level: the
    restatement must reproduce every double bit for bit (same operation order,
    no FMA contraction on either side)."""
    z, meta = ref_ops
    for m in meta:
        key = m["key"]
        d = _desc(m)
        ops = po.Ops(d)
        U, B = z[f"{key}_u"], z[f"{key}_b"]
        for l in range(m["levels"]):
            st = 2 ** l
            u = np.ascontiguousarray(U[::st, ::st]); b = np.ascontiguousarray(B[::st, ::st])
            assert np.array_equal(ops.jacobi(l, u, b), z[f"{key}_jacobi_l{l}"]), (key, l, "jacobi")
            assert np.array_equal(ops.gs_lex(l, u, b), z[f"{key}_gs_l{l}"]), (key, l, "gs")
            r, ss = ops.residual(l, u, b)
            assert np.array_equal(r, z[f"{key}_residual_l{l}"]), (key, l, "residual")
            # Norm() = sqrt(sum r^2 / sum b^2), both sums serial row-major (solvers.hpp:237-276,305-307)
            ref_norm = float(z[f"{key}_residual_norm_l{l}"])
            assert np.sqrt(ss / ops.sumsq(b)) == ref_norm, (key, l, "norm")
            if l < m["levels"] - 1:
                # reference interpolates in place on the aliased vector: fine-only nodes are
                # overwritten, coincident nodes keep the coarse values
                uc = np.ascontiguousarray(U[::2 * st, ::2 * st])
                assert np.array_equal(ops.prolong_overwrite(uc, fine_before=u), z[f"{key}_interp_l{l}"]), (key, l, "interp")


def test_coarse_solver_matches_reference(ref_ops):
    """Solver::Solve (maxit 2000, tol 0.1): same sweep count, same flag, same vector."""
    z, meta = ref_ops
    for m in meta:
        key = m["key"]
        lc = m["levels"] - 1
        st = 2 ** lc
        for smt in (0, 1):
            d = _desc(m, smoother=smt)
            ops = po.Ops(d)
            b = np.ascontiguousarray(z[f"{key}_b"][::st, ::st])
            e, its, flag, rel = ops.coarse_solve(lc, smt, np.zeros_like(b), b)
            norm, sweeps, status = z[f"{key}_coarse_smt{smt}_stats"]
            assert its == int(sweeps) and flag == int(status), (key, smt)
            assert rel == norm, (key, smt)
            assert np.array_equal(e, z[f"{key}_coarse_smt{smt}_e"]), (key, smt)


def test_one_sawtooth_cycle_bit_exact(ref_ops):
    """One SawtoothMGIteration (multigrid.hpp:126-145) from a random state."""
    z, meta = ref_ops
    for m in meta:
        key = m["key"]
        for smt in (0, 1):
            d = _desc(m, smoother=smt)
            s = po.Solver(d)
            s.set_rhs(z[f"{key}_b"]); s.set_solution(z[f"{key}_u"])
            s.cycle()
            assert np.array_equal(s.get_solution(), z[f"{key}_cycle_smt{smt}"]), (key, smt)


with open(os.path.join(G, "ref_solve.json")) as _f:
    SOLVES = json.load(_f)


@pytest.mark.parametrize("case", SOLVES, ids=lambda c: c["key"])
def test_whole_solve_history(case):
    """main.cpp:72-116 end to end. The serial reduction order is the same, so the
    histories agree to the last bit; the tolerance only allows for libm exp/sin
    differences between machines (rhs assembly)."""
    smt = 1 if case["smt"] == 2 else case["smt"]  # -smt 2 runs the Jacobi cycle, main.cpp:103-106
    d = po.make_desc(dim=2, n=case["n"], levels=case["levels"], alpha=case["alpha"],
                     length=case["length"], smoother=smt)
    s = po.Solver(d)
    s.set_rhs(po.fill_rhs_2d(case["n"], case["length"], case["test"]))
    hist, stats = s.solve(1e-11, 1000)
    ref = np.array([float(x) for x in case["hist"]])
    assert len(hist) == len(ref)
    np.testing.assert_allclose(hist, ref, rtol=1e-9)
    refc = np.array([float(x) for x in case["coarse_relres"]])
    np.testing.assert_allclose([st.coarse_relres for st in stats], refc, rtol=2e-5)  # 6 s.d. print
    # sweeps the coarse Solver spent per cycle (counting subclasses of the reference's smoothers)
    assert [st.coarse_iters for st in stats] == case["coarse_counts"]
    ufile = np.load(os.path.join(G, "ref_solve_u.npz"))
    if case["key"] in ufile:
        np.testing.assert_allclose(s.get_solution(), ufile[case["key"]], rtol=1e-9, atol=1e-12)


def _read_hist_file(path):
    vals = [float(x) for x in open(path).read().split()]
    assert int(vals[0]) == len(vals) - 1  # saveVectorOnFile: count line first (utilities.hpp:43-54)
    return np.array(vals[1:])


@pytest.mark.parametrize("fix,n,test,smt", [("web", 145, 1, 1), ("gmgtest", 385, 0, 1)])
def test_reference_committed_fixtures(fix, n, test, smt):
    """The reference's only known-answer files (6 s.d.): MGGS4.txt and x.mtx."""
    d = po.make_desc(dim=2, n=n, levels=5, alpha=1.0, length=10.0, smoother=smt)
    s = po.Solver(d)
    s.set_rhs(po.fill_rhs_2d(n, 10.0, test))
    hist, _ = s.solve(1e-11, 1000)
    ref = _read_hist_file(os.path.join(G, f"fixture_{fix}_MGGS4.txt"))
    assert len(hist) == len(ref)
    np.testing.assert_allclose(hist, ref, rtol=1e-5)
    x = np.load(os.path.join(G, f"fixture_{fix}_x.npz"))["x"]
    np.testing.assert_allclose(s.get_solution().ravel(), x, rtol=2e-5, atol=1e-9)


def test_validation_rejects_bad_grids():
    """The reference reads out of range for n=200, levels=2; we refuse."""
    assert po.lib().orc_validate(po.make_desc(n=200, levels=2)) != 0
    assert po.lib().orc_validate(po.make_desc(n=17, levels=5)) != 0
    assert po.lib().orc_validate(po.make_desc(n=17, levels=4)) == 0


def test_zebra_line_smoother_handles_a_dominant_y_coupling():
    """EXTENSION (no reference counterpart, pinned by convergence theory only): with
    -(dxx + 100 dyy + dzz) the point smoothers leave the y-smooth / x,z-oscillatory error untouched
    (V-cycle factor 0.8-0.9), zebra line Gauss-Seidel along y restores textbook multigrid."""
    facs = {}
    for sm in (po.SMOOTH_RBGS, po.SMOOTH_ZEBRA_Y):
        kw = dict(dim=3, n=33, levels=4, dtype=po.MG_F64, length=1.0, alpha=1.0, cycle=po.CYCLE_V, nu_pre=2, nu_post=2,
                  smoother=sm, omega=1.0, restriction=po.RESTRICT_FULLW, coarse_mode=po.COARSE_FIXED, coarse_maxit=30,
                  outer_pre_gs=0, aniso=(1.0, 100.0, 1.0))
        s = po.Solver(po.make_desc(**kw)); s.set_rhs(po.fill_rhs_3d(33, 1.0, 1.0, 1))
        h, _ = s.solve(1e-13, 6)
        facs[sm] = h[-1] / h[-2]
    assert facs[po.SMOOTH_RBGS] > 0.6 and facs[po.SMOOTH_ZEBRA_Y] < 0.05
    # the same along x (lines on the fast axis) for a dominant x-coupling
    kw.update(aniso=(100.0, 1.0, 1.0), smoother=po.SMOOTH_ZEBRA_X)
    s = po.Solver(po.make_desc(**kw)); s.set_rhs(po.fill_rhs_3d(33, 1.0, 1.0, 1))
    h, _ = s.solve(1e-13, 6)
    assert h[-1] / h[-2] < 0.05
    kw.update(smoother=po.SMOOTH_ZEBRA_Y)   # lines across the strong direction do not help
    s = po.Solver(po.make_desc(**kw)); s.set_rhs(po.fill_rhs_3d(33, 1.0, 1.0, 1))
    h, _ = s.solve(1e-13, 6)
    assert h[-1] / h[-2] > 0.6
    # on the isotropic operator it is simply a stronger smoother than red-black
    kw.update(aniso=(1.0, 1.0, 1.0), smoother=po.SMOOTH_ZEBRA_Y)
    s = po.Solver(po.make_desc(**kw)); s.set_rhs(po.fill_rhs_3d(33, 1.0, 1.0, 1))
    h, _ = s.solve(1e-13, 6)
    assert h[-1] / h[-2] < 0.1


def test_semi_coarsening_and_line_smoother_are_both_needed_for_config5_style_anisotropy():
    """EXTENSION (BASELINE config 5: 'semi-coarsening + line smoother in the strong direction'; no reference
    counterpart): -(dxx + 30 dyy + 0.05 dzz). Point smoothing with standard coarsening stalls, either ingredient
    alone is not enough, both together restore multigrid convergence."""
    fac = {}
    for sm in (po.SMOOTH_RBGS, po.SMOOTH_ZEBRA_Y):
        for semi in (0, 2):
            kw = dict(dim=3, n=65, levels=4, dtype=po.MG_F64, length=1.0, alpha=1.0, cycle=po.CYCLE_V, nu_pre=2, nu_post=2,
                      smoother=sm, omega=1.0, restriction=po.RESTRICT_FULLW, coarse_mode=po.COARSE_FIXED, coarse_maxit=30,
                      outer_pre_gs=0, aniso=(1.0, 30.0, 0.05), semi_xy=semi)
            s = po.Solver(po.make_desc(**kw)); s.set_rhs(po.fill_rhs_3d(65, 1.0, 1.0, 1))
            h, _ = s.solve(1e-13, 6)
            fac[(sm, semi)] = h[-1] / h[-2]
    assert fac[(po.SMOOTH_RBGS, 0)] > 0.8 and fac[(po.SMOOTH_RBGS, 2)] > 0.6 and fac[(po.SMOOTH_ZEBRA_Y, 0)] > 0.4
    assert fac[(po.SMOOTH_ZEBRA_Y, 2)] < 0.05
