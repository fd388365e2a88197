"""GPU parity: every HIP kernel and the whole cycle, through the C-ABI (libmg_hip.so),
against the CPU oracle on the same seeded inputs.

Bars: single operators are BIT-EXACT in fp64 and fp32 (same operation
order, no FMA contraction, IEEE division); sums of squares differ only by summation
order (rtol 1e-12 fp64); whole solves follow the reference history to rtol 1e-6 (the
coarse iterate-to-tolerance loop may flip by one sweep when a norm differs in the last
bit).
"""
import json
import os

import numpy as np
import pytest

from multigrid_prj_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def pair(**kw):
    """(HIP solver, oracle Ops, oracle desc) for identical parameters."""
    dg = capi.make_desc(**kw)
    do = po.make_desc(**kw)
    return capi.Solver(dg), po.Ops(do), do


def rnd(rng, shape, dtype):
    return rng.standard_normal(shape).astype(dtype)


CASES = [
    dict(dim=2, n=33, levels=3, dtype=capi.MG_F64, alpha=1.0, length=10.0),
    dict(dim=2, n=25, levels=2, dtype=capi.MG_F64, alpha=0.7, length=1.0),
    dict(dim=2, n=145, levels=5, dtype=capi.MG_F64, alpha=1.0, length=10.0),
    dict(dim=2, n=33, levels=3, dtype=capi.MG_F32, alpha=1.0, length=10.0),
    dict(dim=3, n=17, levels=3, dtype=capi.MG_F64, alpha=1.0, length=1.0),
    dict(dim=3, n=33, levels=2, dtype=capi.MG_F64, alpha=2.5, length=4.0),
    dict(dim=3, n=21, levels=3, dtype=capi.MG_F32, alpha=1.0, length=1.0),
    dict(dim=3, n=67, levels=2, dtype=capi.MG_F64, alpha=1.0, length=1.0),
    dict(dim=3, n=129, levels=4, dtype=capi.MG_F64, alpha=1.0, length=1.0),
    dict(dim=3, n=131, levels=2, dtype=capi.MG_F32, alpha=1.0, length=1.0),
    # fast-path shapes (mg_jacobi_fast.hip): n % 4 == 1 in fp32, odd n in fp64
    dict(dim=3, n=65, levels=3, dtype=capi.MG_F32, alpha=1.0, length=1.0),
    dict(dim=3, n=129, levels=2, dtype=capi.MG_F32, alpha=3.0, length=2.0),
    dict(dim=3, n=97, levels=2, dtype=capi.MG_F64, alpha=1.0, length=1.0),
    # anisotropic operator -(dxx + dyy + eps dzz) with semi-coarsening (BASELINE config 5 in miniature)
    # semi_xy = k: the first k transitions coarsen x,y only, the rest all three axes
    dict(dim=3, n=33, levels=3, dtype=capi.MG_F64, alpha=1.0, length=1.0, semi_xy=2, aniso=(1.0, 1.0, 0.01)),
    dict(dim=3, n=65, levels=4, dtype=capi.MG_F32, alpha=2.0, length=1.0, semi_xy=1, aniso=(1.0, 0.5, 0.1)),
]
IDS = [f"{c['dim']}d-n{c['n']}-L{c['levels']}-{'f64' if c['dtype'] == 0 else 'f32'}{'-semi' if c.get('semi_xy') else ''}" for c in CASES]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_smoothers_bit_exact(case):
    rng = np.random.default_rng(1)
    for omega in (1.0, 6.0 / 7.0):
        s, ops, do = pair(omega=omega, **case)
        with s:
            for l in range(case["levels"]):
                shp = s.level_shape(l)
                u, b = rnd(rng, shp, s.np), rnd(rng, shp, s.np)
                s.set_array(capi.ARR_E, l, u); s.set_array(capi.ARR_RHS, l, b)
                s.smooth(l, capi.SMOOTH_JACOBI, 1, capi.ARR_E, capi.ARR_RHS)
                assert np.array_equal(s.get_array(capi.ARR_E, l), ops.jacobi(l, u, b)), ("jacobi", l, omega)
                s.smooth(l, capi.SMOOTH_JACOBI, 2, capi.ARR_E, capi.ARR_RHS)
                assert np.array_equal(s.get_array(capi.ARR_E, l),
                                      ops.smooth(l, po.SMOOTH_JACOBI, 3, u, b)), ("jacobi x3", l, omega)
                if omega != 1.0:
                    continue
                s.set_array(capi.ARR_E, l, u)
                s.smooth(l, capi.SMOOTH_RBGS, 2, capi.ARR_E, capi.ARR_RHS)
                assert np.array_equal(s.get_array(capi.ARR_E, l), ops.smooth(l, po.SMOOTH_RBGS, 2, u, b)), ("rbgs", l)
                if np.prod(shp) <= 70 ** 3:  # one-workgroup wavefront kernel: keep it small
                    s.set_array(capi.ARR_E, l, u)
                    s.smooth(l, capi.SMOOTH_GS_LEX, 2, capi.ARR_E, capi.ARR_RHS)
                    assert np.array_equal(s.get_array(capi.ARR_E, l),
                                          ops.smooth(l, po.SMOOTH_GS_LEX, 2, u, b)), ("gs_lex", l)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_residual_and_norms(case):
    rng = np.random.default_rng(2)
    s, ops, do = pair(**case)
    rtol = 1e-12 if case["dtype"] == capi.MG_F64 else 1e-6
    with s:
        for l in range(case["levels"]):
            shp = s.level_shape(l)
            u, b = rnd(rng, shp, s.np), rnd(rng, shp, s.np)
            s.set_array(capi.ARR_E, l, u); s.set_array(capi.ARR_RHS, l, b)
            ss = s.residual(l, capi.ARR_E, capi.ARR_RHS, capi.ARR_TMP)
            r_ref, ss_ref = ops.residual(l, u, b)
            assert np.array_equal(s.get_array(capi.ARR_TMP, l), r_ref), ("residual", l)
            assert ss == pytest.approx(ss_ref, rel=rtol)
            assert s.residual(l, capi.ARR_E, capi.ARR_RHS, -1) == ss  # non-saving branch, same reduction
            assert s.sumsq(l, capi.ARR_RHS) == pytest.approx(ops.sumsq(b), rel=rtol)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_transfers_and_correction_bit_exact(case):
    rng = np.random.default_rng(3)
    s, ops, do = pair(**case)
    with s:
        for l in range(case["levels"] - 1):
            fine = rnd(rng, s.level_shape(l), s.np)
            coarse = rnd(rng, s.level_shape(l + 1), s.np)
            s.set_array(capi.ARR_E, l, fine)
            s.restrict(l, capi.RESTRICT_INJECT, capi.ARR_E, capi.ARR_RHS)
            assert np.array_equal(s.get_array(capi.ARR_RHS, l + 1), ops.inject(fine)), ("inject", l)
            s.restrict(l, capi.RESTRICT_FULLW, capi.ARR_E, capi.ARR_RHS)
            assert np.array_equal(s.get_array(capi.ARR_RHS, l + 1), ops.restrict_fw(fine)), ("fullw", l)
            s.set_array(capi.ARR_E, l + 1, coarse)
            s.prolong(l + 1, False, capi.ARR_E, capi.ARR_E)
            assert np.array_equal(s.get_array(capi.ARR_E, l), ops.prolong_overwrite(coarse)), ("prolong", l)
            s.set_array(capi.ARR_U, l, fine); s.set_array(capi.ARR_U, l + 1, coarse)
            s.prolong(l + 1, True, capi.ARR_U, capi.ARR_U)
            assert np.array_equal(s.get_array(capi.ARR_U, l), ops.prolong_add(coarse, fine)), ("prolong_add", l)
        u, e = rnd(rng, s.level_shape(0), s.np), rnd(rng, s.level_shape(0), s.np)
        s.set_array(capi.ARR_U, 0, u); s.set_array(capi.ARR_E, 0, e)
        s.correct()
        u2, e2 = ops.correct(u, e)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), u2)
        assert not s.get_array(capi.ARR_E, 0).any()


@pytest.mark.parametrize("case", [c for c in CASES if c["n"] <= 67], ids=[i for c, i in zip(CASES, IDS) if c["n"] <= 67])
@pytest.mark.parametrize("smoother", [capi.SMOOTH_GS_LEX, capi.SMOOTH_JACOBI, capi.SMOOTH_RBGS])
def test_coarse_solver(case, smoother):
    """Solver::Solve in one persistent workgroup: same sweep count, flag and vector."""
    rng = np.random.default_rng(4)
    s, ops, do = pair(smoother=smoother, **case)
    lc = case["levels"] - 1
    with s:
        b = rnd(rng, s.level_shape(lc), s.np)
        s.set_array(capi.ARR_RHS, lc, b); s.zero_array(capi.ARR_E, lc)
        st = s.coarse_solve(lc, capi.ARR_E, capi.ARR_RHS)
        e, its, flag, rel = ops.coarse_solve(lc, smoother, np.zeros_like(b), b)
        assert (st.coarse_iters, st.coarse_flag) == (its, flag)
        assert st.coarse_relres == pytest.approx(rel, rel=1e-10 if case["dtype"] == 0 else 1e-5)
        assert np.array_equal(s.get_array(capi.ARR_E, lc), e)
    # fixed-sweep mode (extension)
    s, ops, do = pair(smoother=smoother, coarse_mode=capi.COARSE_FIXED, coarse_maxit=7, **case)
    with s:
        s.set_array(capi.ARR_RHS, lc, b); s.zero_array(capi.ARR_E, lc)
        st = s.coarse_solve(lc, capi.ARR_E, capi.ARR_RHS)
        e, its, flag, rel = ops.coarse_solve(lc, smoother, np.zeros_like(b), b, maxit=7, fixed=True)
        assert st.coarse_iters == 7 and np.array_equal(s.get_array(capi.ARR_E, lc), e)


@pytest.mark.parametrize("case", [c for c in CASES if c["n"] <= 67], ids=[i for c, i in zip(CASES, IDS) if c["n"] <= 67])
@pytest.mark.parametrize("smoother", [capi.SMOOTH_GS_LEX, capi.SMOOTH_JACOBI])
def test_one_sawtooth_cycle(case, smoother):
    """SawtoothMGIteration::apply_iteration_to_vec from a random state: bit-exact when the
    coarse solver spends the same number of sweeps (it does unless a norm ties at 0.1)."""
    rng = np.random.default_rng(5)
    kw = dict(smoother=smoother, **case)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    with sg:
        u, b = rnd(rng, sg.level_shape(0), sg.np), rnd(rng, sg.level_shape(0), sg.np)
        sg.set_solution(u); sg.set_rhs(b); so.set_solution(u); so.set_rhs(b)
        st_g, st_o = sg.cycle(), so.cycle()
        assert st_g.coarse_iters == st_o.coarse_iters and st_g.coarse_flag == st_o.coarse_flag
        assert st_g.fine_sumsq_r == pytest.approx(st_o.fine_sumsq_r, rel=1e-12 if case["dtype"] == 0 else 1e-6)
        assert np.array_equal(sg.get_solution(), so.get_solution())
        assert np.array_equal(sg.get_array(capi.ARR_RES, 0), so.get_residual())


VC = [
    dict(dim=3, n=33, levels=3, dtype=capi.MG_F64, smoother=capi.SMOOTH_JACOBI, omega=6 / 7),
    dict(dim=3, n=33, levels=3, dtype=capi.MG_F64, smoother=capi.SMOOTH_RBGS, omega=1.0),
    dict(dim=3, n=65, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW),
    dict(dim=3, n=33, levels=3, dtype=capi.MG_F32, smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW),
    dict(dim=2, n=65, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_RBGS, omega=1.0, restriction=capi.RESTRICT_FULLW),
    # grids wide enough for the fused kernels (double sweep, residual+restriction, and the
    # prolongation folded into the post-smoothing pair) on one, two and two levels respectively
    dict(dim=3, n=129, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW),
    dict(dim=3, n=257, levels=5, dtype=capi.MG_F64, smoother=capi.SMOOTH_JACOBI, omega=0.8, restriction=capi.RESTRICT_INJECT),
    dict(dim=3, n=257, levels=5, dtype=capi.MG_F32, smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW),
    # rows of 192 vectors (n = 385 = 3 * 128 + 1): fused pair, residual + restriction and the folded prolongation between 385^3 and 193^3
    dict(dim=3, n=385, levels=3, dtype=capi.MG_F64, smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW),
    dict(dim=3, n=385, levels=3, dtype=capi.MG_F64, smoother=capi.SMOOTH_RBGS, omega=1.0, restriction=capi.RESTRICT_FULLW),
    # zebra line smoother along y on an operator whose y-coupling dominates (point smoothers stall at 0.8-0.9)
    dict(dim=3, n=65, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_ZEBRA_Y, omega=1.0, restriction=capi.RESTRICT_FULLW,
         aniso=(1.0, 100.0, 1.0)),
    dict(dim=2, n=129, levels=5, dtype=capi.MG_F64, smoother=capi.SMOOTH_ZEBRA_Y, omega=1.0, restriction=capi.RESTRICT_FULLW,
         aniso=(1.0, 100.0, 1.0)),
    dict(dim=3, n=65, levels=4, dtype=capi.MG_F32, smoother=capi.SMOOTH_ZEBRA_Y, omega=1.0, restriction=capi.RESTRICT_FULLW,
         aniso=(1.0, 30.0, 1.0)),
    # zebra lines along x (the fast axis: LDS-staged chunks) on a dominant x-coupling
    dict(dim=3, n=65, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_ZEBRA_X, omega=1.0, restriction=capi.RESTRICT_FULLW,
         aniso=(100.0, 1.0, 1.0)),
    dict(dim=2, n=129, levels=5, dtype=capi.MG_F64, smoother=capi.SMOOTH_ZEBRA_X, omega=1.0, restriction=capi.RESTRICT_FULLW,
         aniso=(100.0, 1.0, 1.0)),
    dict(dim=3, n=129, levels=5, dtype=capi.MG_F32, smoother=capi.SMOOTH_ZEBRA_X, omega=1.0, restriction=capi.RESTRICT_FULLW,
         semi_xy=2, aniso=(30.0, 1.0, 0.05)),
    # BASELINE config 5 as worded -- semi-coarsening + a line smoother in the strong direction: weak z-coupling (eps = 0.05,
    # log4(1/eps) ~ 2 semi-coarsenings) and a dominant y-coupling (zebra lines along y); neither ingredient alone converges
    dict(dim=3, n=65, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_ZEBRA_Y, omega=1.0, restriction=capi.RESTRICT_FULLW,
         semi_xy=2, aniso=(1.0, 30.0, 0.05)),
    dict(dim=3, n=129, levels=5, dtype=capi.MG_F32, smoother=capi.SMOOTH_ZEBRA_Y, omega=1.0, restriction=capi.RESTRICT_FULLW,
         semi_xy=2, aniso=(1.0, 30.0, 0.05)),
    # red-black with the fused one-pass sweep and the prolongation folded into the first post-sweep
    dict(dim=3, n=129, levels=4, dtype=capi.MG_F64, smoother=capi.SMOOTH_RBGS, omega=1.0, restriction=capi.RESTRICT_FULLW),
    dict(dim=3, n=257, levels=5, dtype=capi.MG_F32, smoother=capi.SMOOTH_RBGS, omega=1.0, restriction=capi.RESTRICT_FULLW),
    # anisotropic eps with k ~ log4(1/eps) semi-coarsenings followed by standard ones:
    # eps = 0.01, k = 3 (coarsest 9 x 3 x 3... one-workgroup solve) and eps = 0.25, k = 1 with a coarsest
    # grid too big for one workgroup (33 x 33 x 65: swept with the regular kernels)
    dict(dim=3, n=65, levels=6, dtype=capi.MG_F64, smoother=capi.SMOOTH_JACOBI, omega=0.8, restriction=capi.RESTRICT_FULLW,
         semi_xy=3, aniso=(1.0, 1.0, 0.01)),
    dict(dim=3, n=129, levels=3, dtype=capi.MG_F64, smoother=capi.SMOOTH_RBGS, omega=1.0, restriction=capi.RESTRICT_FULLW,
         semi_xy=1, aniso=(1.0, 1.0, 0.25)),
]


@pytest.mark.parametrize("case", VC, ids=lambda c: f"{c['dim']}d-n{c['n']}-s{c['smoother']}-r{c.get('restriction', 0)}-t{c['dtype']}{'-semi' if c.get('semi_xy') else ''}")
def test_vcycle_extension(case):
    """V(2,2) with fixed coarse sweeps (BASELINE configs 2-4 in miniature): the cycle is
    a fixed sequence of bit-exact kernels, so the solution matches bit for bit."""
    kw = dict(length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              coarse_mode=capi.COARSE_FIXED, coarse_maxit=30, outer_pre_gs=0, **case)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    n = case["n"]
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1) if case["dim"] == 3 else po.fill_rhs_2d(n, 1.0, 2)
    with sg:
        sg.set_rhs(b); so.set_rhs(b)
        for _ in range(3):
            sg.cycle(); so.cycle()
        assert np.array_equal(sg.get_solution(), so.get_solution())
        hg, _ = sg.solve(1e-9, 4); ho, _ = so.solve(1e-9, 4)
        np.testing.assert_allclose(hg, ho, rtol=1e-10 if case["dtype"] == 0 else 1e-4)
        # Injection right after a red-black sweep aliases (the residual vanishes on the last
        # colour), a textbook failure both implementations reproduce; every other pairing
        # must converge like a multigrid cycle.
        # (The 33 x 33 x 65 coarsest grid of the n=129 semi case is only swept 30 times, far from
        # solved: it checks the swept-coarse-level path bit for bit, not its convergence.)
        # (likewise the 97^3 coarsest grid of the n = 385 cases)
        swept_coarse = (case.get("semi_xy") and case["n"] == 129) or case["n"] == 385
        at_floor = case["dtype"] == capi.MG_F32 and hg[0] < 1e-5   # fp32 round-off floor reached within the first cycles
        if not (case["smoother"] == capi.SMOOTH_RBGS and case.get("restriction", 0) == capi.RESTRICT_INJECT) and not swept_coarse \
                and not at_floor:
            assert hg[-1] < 0.2 * hg[0]


@pytest.mark.parametrize("smoother", [capi.SMOOTH_JACOBI, capi.SMOOTH_RBGS])
def test_vcycle_headline_size_513_bit_exact(smoother):
    """The benchmarked configuration itself (513^3 fp64, 6 levels, V(2,2), full weighting, 17^3 coarse
    grid iterated to 0.1): two cycles on the GPU -- fused sweep pairs / one-pass red-black sweeps with
    256-lane rows, residual+restriction fused, prolongation folded into the post-smoothing, LDS coarse
    solver -- against two cycles of the oracle, every one of the 1.35e8 unknowns bit for bit."""
    n = 513
    kw = dict(dim=3, n=n, levels=6, dtype=capi.MG_F64, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              smoother=smoother, omega=6 / 7 if smoother == capi.SMOOTH_JACOBI else 1.0,
              restriction=capi.RESTRICT_FULLW, coarse_mode=capi.COARSE_TOL, coarse_tol=0.1, coarse_maxit=2000, outer_pre_gs=0)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1)
    with sg:
        sg.set_rhs(b); so.set_rhs(b)
        del b
        for _ in range(2):
            stg = sg.cycle(); sto = so.cycle()
            assert stg.coarse_iters == sto.coarse_iters
        assert np.array_equal(sg.get_solution(), so.get_solution())


@pytest.mark.parametrize("n,dtype,levels,smoother", [(257, capi.MG_F64, 5, capi.SMOOTH_JACOBI), (513, capi.MG_F32, 6, capi.SMOOTH_JACOBI),
                                                    (257, capi.MG_F64, 5, capi.SMOOTH_RBGS), (513, capi.MG_F32, 6, capi.SMOOTH_RBGS)],
                         ids=["257-f64-jacobi", "513-f32-jacobi", "257-f64-redblack", "513-f32-redblack"])
def test_solve_takes_the_residual_norm_inside_the_next_pre_smoothing_pair(n, dtype, levels, smoother):
    """mg_solve on a level wide enough for the wide-tile pair (mg_pair_wide.hip: NORM) computes each history entry inside the
    first pre-smoothing pair of the NEXT cycle and drops the speculative pair when the loop stops (src/main.cpp:86-89). The
    outer loop done by hand -- mg_cycle + mg_residual per iteration, the separate norm kernel -- must give the same history
    (summation order differs: rtol), the same number of entries when the tolerance stops the loop, and the same iterate BIT
    FOR BIT; a cycle after the solve must continue from that iterate. Red-black: the FIRST pre-smoothing sweep carries the
    norm (one out-of-place launch, dropped on stop) and the cycle runs the second."""
    kw = dict(dim=3, n=n, levels=levels, dtype=dtype, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              smoother=smoother, omega=6 / 7 if smoother == capi.SMOOTH_JACOBI else 1.0, restriction=capi.RESTRICT_FULLW,
              coarse_mode=capi.COARSE_FIXED, coarse_maxit=20, outer_pre_gs=0)
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1)
    if dtype == capi.MG_F32:
        b = b.astype(np.float32)
    rtol = 1e-12 if dtype == capi.MG_F64 else 1e-5
    with capi.Solver(capi.make_desc(**kw)) as s1, capi.Solver(capi.make_desc(**kw)) as s2:
        s1.set_rhs(b); s2.set_rhs(b)
        nb = s2.sumsq(0, capi.ARR_RHS)
        hand = [np.sqrt(s2.residual(0, capi.ARR_U, capi.ARR_RHS, -1) / nb)]
        hand_iters = []
        for _ in range(3):
            hand_iters.append(s2.cycle().coarse_iters)
            hand.append(np.sqrt(s2.residual(0, capi.ARR_U, capi.ARR_RHS, -1) / nb))
        hist, stats = s1.solve(0.0, 3)
        assert len(hist) == 4 and [st.coarse_iters for st in stats] == hand_iters
        np.testing.assert_allclose(hist, hand, rtol=rtol)
        assert np.array_equal(s1.get_solution(), s2.get_solution())
        # the tolerance stops the loop after the same cycle
        tol = float(np.sqrt(hand[1] * hand[2]))
        s1.zero_array(capi.ARR_U, 0); s2.zero_array(capi.ARR_U, 0)
        hist, _ = s1.solve(tol, 10)
        assert len(hist) == 3 and hist[-1] <= tol < hist[-2]
        np.testing.assert_allclose(hist, hand[:3], rtol=rtol)
        for _ in range(2):
            s2.cycle()
        assert np.array_equal(s1.get_solution(), s2.get_solution())
        # and the solver carries on from there
        s1.cycle(); s2.cycle()
        assert np.array_equal(s1.get_solution(), s2.get_solution())


@pytest.mark.parametrize("smoother", [capi.SMOOTH_JACOBI, capi.SMOOTH_RBGS])
def test_config5_anisotropic_semi_coarsened_513_bit_exact(smoother):
    """BASELINE config 5 at full size: -(dxx + dyy + 0.01 dzz) on 513^3, 8 levels of which the first three
    coarsen x,y only (k = log4(1/eps) semi-coarsenings, then standard ones), V(2,2), full weighting, 20 coarse
    sweeps: two cycles on the GPU against two cycles of the oracle, all 1.35e8 unknowns bit for bit, and the
    cycle must converge like multigrid (standard coarsening stalls at 0.9 on this operator)."""
    n = 513
    kw = dict(dim=3, n=n, levels=8, dtype=capi.MG_F64, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              smoother=smoother, omega=6 / 7 if smoother == capi.SMOOTH_JACOBI else 1.0, restriction=capi.RESTRICT_FULLW,
              coarse_mode=capi.COARSE_FIXED, coarse_maxit=20, outer_pre_gs=0, aniso=(1.0, 1.0, 0.01), semi_xy=3)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1)
    with sg:
        assert [sg.level_shape(l) for l in (0, 3, 4, 7)] == [(513, 513, 513), (513, 65, 65), (257, 33, 33), (33, 5, 5)]
        sg.set_rhs(b); so.set_rhs(b)
        del b
        for _ in range(2):
            sg.cycle(); so.cycle()
        assert np.array_equal(sg.get_solution(), so.get_solution())
        hist, _ = sg.solve(0.0, 3)
        assert hist[-1] / hist[-2] < (0.3 if smoother == capi.SMOOTH_JACOBI else 0.15), hist


def test_config4_grid_1025_fp32_fused_operators_bit_exact():
    """BASELINE config 4's grid (1025^3 fp32, rows of 256 four-float vectors) through every fused finest-level
    kernel -- the sweep pair, residual + full weighting, and the pair that folds the prolongation -- as ONE
    two-level V(2,2) cycle (the 513^3 'coarse' level is swept 4 times with the same fused pair) against the
    oracle on the whole grid, 1.08e9 unknowns bit for bit. (8 ranks of the distributed run each hold a slab
    of exactly this grid; k ranks == 1 rank is tests/test_distributed.py's part.)"""
    n = 1025
    kw = dict(dim=3, n=n, levels=2, dtype=capi.MG_F32, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW,
              coarse_mode=capi.COARSE_FIXED, coarse_maxit=4, outer_pre_gs=0)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1).astype(np.float32)
    with sg:
        sg.set_rhs(b); so.set_rhs(b)
        del b
        sg.cycle(); so.cycle()
        ug = sg.get_solution(); uo = so.get_solution()
        assert np.array_equal(ug, uo)
        del ug, uo
        # the finest-level residual through the vectorised path, and its norm
        ss = sg.residual(0, capi.ARR_U, capi.ARR_RHS, -1)
        assert ss == pytest.approx(so.residual_fine(), rel=1e-5)
    so.close()


@pytest.mark.parametrize("n,dtype", [(513, "f64"), (513, "f32")])
def test_wide_tile_kernels_equal_the_row_column_kernels_on_every_piece_geometry(n, dtype):
    """tools/pairbench (built by __graft_entry__.build()): every variant of the wide-tile kernels -- plain / zero-guess /
    prolongation-folding Jacobi pairs, one-pass red-black sweeps, residual + full weighting -- on a whole level and on the
    pieces of a z-slab (whole slab, interior planes, the two boundary pieces in one launch; 64 and 32 planes, coarse planes
    addressed by global index) through the product's launchers, against the round-2 kernels that the parity tests pinned to
    the oracle: every 32-bit word of the output arrays, ghost planes and padding included."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = os.path.join(root, "tools", "pairbench")
    if not os.path.exists(exe):
        pytest.skip("tools/pairbench not built (python __graft_entry__.py build)")
    p = subprocess.run([exe, str(n), "2", dtype, "all"], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0 and "all variants bit-equal" in p.stdout and "MISMATCH" not in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


_RANGES_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from multigrid_prj_amd import capi
from oracle import pyoracle as po
n, dtype = int(sys.argv[2]), int(sys.argv[3])
kw = dict(dim=3, n=n, levels=4, dtype=dtype, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
          smoother=int(sys.argv[4]), omega=6 / 7 if int(sys.argv[4]) == capi.SMOOTH_JACOBI else 1.0, restriction=capi.RESTRICT_FULLW,
          coarse_mode=capi.COARSE_FIXED, coarse_maxit=20, outer_pre_gs=0)
b = po.fill_rhs_3d(n, 1.0, 1.0, 1)
if dtype == capi.MG_F32:
    b = b.astype(np.float32)
sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
with sg:
    sg.set_rhs(b); so.set_rhs(b)
    for _ in range(2):
        sg.cycle(); so.cycle()
    assert np.array_equal(sg.get_solution(), so.get_solution())
print("ranges ok")
"""


@pytest.mark.parametrize("n,dtype,smoother", [(257, capi.MG_F64, capi.SMOOTH_JACOBI), (513, capi.MG_F32, capi.SMOOTH_RBGS)])
def test_wide_tile_kernels_dealt_as_balanced_ranges_keep_the_bits(n, dtype, smoother):
    """The wide-tile kernels (k_pairw, k_rrw) can deal their work as balanced plane RANGES instead of z-chunks (MG_PW_MODE=0,
    MG_RRW_MODE=0: a workgroup then changes tile in mid-launch and chunk boundaries fall anywhere). The switch is read once per
    process, so a child process runs two V(2,2) cycles that way against the oracle, bit for bit."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, MG_PW_MODE="0", MG_RRW_MODE="0")
    p = subprocess.run([sys.executable, "-c", _RANGES_SCRIPT, root, str(n), str(dtype), str(smoother)], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ranges ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_config4_as_worded_1025_fp32_seven_level_vcycles_bit_exact():
    """BASELINE config 4's hierarchy on one GPU: 3-D Poisson 1025^3 fp32, SEVEN levels (1025 ... 17), V(2,2) Jacobi omega = 6/7,
    full weighting, 17^3 coarse grid iterated to relative residual 0.1 -- two whole cycles against the oracle, all 1.08e9
    unknowns bit for bit, the coarse solver's sweep counts equal. (What 8 ranks add to it -- slabs and exchanges -- is
    tests/test_distributed.py's part: k ranks == 1 rank.)"""
    n = 1025
    kw = dict(dim=3, n=n, levels=7, dtype=capi.MG_F32, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              smoother=capi.SMOOTH_JACOBI, omega=6 / 7, restriction=capi.RESTRICT_FULLW,
              coarse_mode=capi.COARSE_TOL, coarse_tol=0.1, coarse_maxit=2000, outer_pre_gs=0)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1).astype(np.float32)
    with sg:
        assert [sg.level_shape(l)[0] for l in range(7)] == [1025, 513, 257, 129, 65, 33, 17]
        sg.set_rhs(b); so.set_rhs(b)
        del b
        for _ in range(2):
            stg = sg.cycle(); sto = so.cycle()
            assert stg.coarse_iters == sto.coarse_iters
        ug = sg.get_solution(); uo = so.get_solution()
        assert np.array_equal(ug, uo)
        del ug, uo
        # and it is a multigrid cycle: the residual falls by the headline configuration's factor (0.23 per cycle in fp64)
        hist, _ = sg.solve(0.0, 2)
        assert hist[2] < 0.4 * hist[1], hist
    so.close()


@pytest.mark.parametrize("zsm", [capi.SMOOTH_ZEBRA_Y, capi.SMOOTH_ZEBRA_X], ids=["y-lines", "x-lines"])
def test_config5_as_worded_semi_coarsening_plus_line_smoother_513_bit_exact(zsm):
    """BASELINE config 5 AS WORDED at full size: anisotropic Poisson 513^3, eps = 0.01 in z, semi-coarsening (the first
    three transitions coarsen x,y only, 8 levels) AND a zebra line smoother -- lines along y (or x): neither crosses the
    z-slabs. One V(2,2) cycle on the GPU against one cycle of the oracle, all 1.35e8 unknowns bit for bit, then the
    reduction factor of further cycles (point smoothers on this hierarchy: 0.21 Jacobi / 0.067 red-black per cycle)."""
    n = 513
    kw = dict(dim=3, n=n, levels=8, dtype=capi.MG_F64, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
              smoother=zsm, omega=1.0, restriction=capi.RESTRICT_FULLW,
              coarse_mode=capi.COARSE_FIXED, coarse_maxit=20, outer_pre_gs=0, aniso=(1.0, 1.0, 0.01), semi_xy=3)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    b = po.fill_rhs_3d(n, 1.0, 1.0, 1)
    with sg:
        assert [sg.level_shape(l) for l in (0, 3, 4, 7)] == [(513, 513, 513), (513, 65, 65), (257, 33, 33), (33, 5, 5)]
        sg.set_rhs(b); so.set_rhs(b)
        del b
        sg.cycle(); so.cycle()
        assert np.array_equal(sg.get_solution(), so.get_solution())
        hist, _ = sg.solve(0.0, 3)
        assert hist[-1] / hist[-2] < 0.15, hist
    so.close()


with open(os.path.join(G, "ref_solve.json")) as _f:
    SOLVES = json.load(_f)


@pytest.mark.parametrize("case", SOLVES, ids=lambda c: c["key"])
def test_whole_solve_vs_reference_golden(case):
    """End to end `Multigrid -n … -smt …` on the GPU against the REAL reference's history
    (tests/golden/ref_solve.json, generated from the compiled reference)."""
    smt = 1 if case["smt"] == 2 else case["smt"]
    kw = dict(dim=2, n=case["n"], levels=case["levels"], alpha=case["alpha"], length=case["length"], smoother=smt)
    ref = np.array([float(x) for x in case["hist"]])
    with capi.Solver(capi.make_desc(**kw)) as s:
        s.set_rhs(po.fill_rhs_2d(case["n"], case["length"], case["test"]))
        hist, stats = s.solve(1e-11, 1000)
        assert abs(len(hist) - len(ref)) <= 1
        m = min(len(hist), len(ref))
        np.testing.assert_allclose(hist[:m], ref[:m], rtol=2e-3)   # free-running: coarse count may flip
        np.testing.assert_allclose(hist[:4], ref[:4], rtol=1e-9)   # early cycles are far from the tie
        ufile = np.load(os.path.join(G, "ref_solve_u.npz"))
        if case["key"] in ufile:
            np.testing.assert_allclose(s.get_solution(), ufile[case["key"]], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("case", SOLVES, ids=lambda c: c["key"])
def test_whole_solve_lockstep_vs_reference_golden(case):
    """Lock-step parity: the coarse Solver's stop test `Norm() > 0.1` is the one chaotic
    element of the reference algorithm, so mg_solve_lockstep replays the sweep counts the REAL reference
    spent per cycle (golden `coarse_counts`) and everything else is held tightly: the same number of
    outer iterations, every history entry to rtol 1e-9 (norms differ from the serial loop by summation
    order only), BASELINE config 1's final solution vector to rtol 1e-9 against the reference's and bit
    for bit against the oracle (north_star: 'final solution vector and per-cycle residual')."""
    smt = 1 if case["smt"] == 2 else case["smt"]
    kw = dict(dim=2, n=case["n"], levels=case["levels"], alpha=case["alpha"], length=case["length"], smoother=smt)
    ref = np.array([float(x) for x in case["hist"]])
    b = po.fill_rhs_2d(case["n"], case["length"], case["test"])
    with capi.Solver(capi.make_desc(**kw)) as s:
        s.set_rhs(b)
        hist, stats = s.solve_lockstep(case["coarse_counts"], 1e-11, 1000)
        assert len(hist) == len(ref)
        np.testing.assert_allclose(hist, ref, rtol=1e-9)
        assert [st.coarse_iters for st in stats] == case["coarse_counts"]
        refc = np.array([float(x) for x in case["coarse_relres"]])
        np.testing.assert_allclose([st.coarse_relres for st in stats], refc, rtol=2e-5)  # the reference prints 6 digits
        u = s.get_solution()
    ufile = np.load(os.path.join(G, "ref_solve_u.npz"))
    if case["key"] in ufile:      # every n <= 65 case and BASELINE config 1 (n = 257, both smoothers)
        np.testing.assert_allclose(u, ufile[case["key"]], rtol=1e-9, atol=1e-12)
    o = po.Solver(po.make_desc(**kw)); o.set_rhs(b)
    ho, so = o.solve(1e-11, 1000)
    assert [st.coarse_iters for st in so] == case["coarse_counts"]
    assert np.array_equal(u, o.get_solution())


@pytest.mark.parametrize("fix,n,test", [("web", 145, 1), ("gmgtest", 385, 0)])
def test_reference_committed_fixtures_gpu(fix, n, test):
    """The reference's own MGGS4.txt / x.mtx (6 s.d.)."""
    with capi.Solver(capi.make_desc(dim=2, n=n, levels=5, alpha=1.0, length=10.0, smoother=capi.SMOOTH_JACOBI)) as s:
        s.set_rhs(po.fill_rhs_2d(n, 10.0, test))
        hist, _ = s.solve(1e-11, 1000)
        vals = [float(x) for x in open(os.path.join(G, f"fixture_{fix}_MGGS4.txt")).read().split()]
        ref = np.array(vals[1:])
        assert len(hist) == len(ref)
        np.testing.assert_allclose(hist, ref, rtol=2e-3)
        x = np.load(os.path.join(G, f"fixture_{fix}_x.npz"))["x"]
        np.testing.assert_allclose(s.get_solution().ravel(), x, rtol=2e-5, atol=1e-8)


def test_full_size_jacobi_and_residual_513():
    """BASELINE full size (513^3 fp64, the headline grid): one sweep and one residual,
    bit-compared against the oracle on the whole 1.35e8-point grid."""
    n = 513
    kw = dict(dim=3, n=n, levels=6, dtype=capi.MG_F64, length=1.0, alpha=1.0, omega=1.0)
    s, ops, do = pair(**kw)
    rng = np.random.default_rng(7)
    with s:
        u = rng.random((n, n, n)); b = rng.random((n, n, n))
        s.set_array(capi.ARR_U, 0, u); s.set_array(capi.ARR_RHS, 0, b)
        s.smooth(0, capi.SMOOTH_JACOBI, 1, capi.ARR_U, capi.ARR_RHS)
        ref = ops.jacobi(0, u, b)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ref)
        ss = s.residual(0, capi.ARR_U, capi.ARR_RHS, capi.ARR_TMP)
        r_ref, ss_ref = ops.residual(0, ref, b)
        assert np.array_equal(s.get_array(capi.ARR_TMP, 0), r_ref)
        assert ss == pytest.approx(ss_ref, rel=1e-11)


@pytest.mark.parametrize("n,dtype,omega", [(129, capi.MG_F64, 6 / 7), (257, capi.MG_F64, 1.0), (257, capi.MG_F32, 6 / 7),
                                           (513, capi.MG_F64, 6 / 7),
                                           # rows of 192 / 384 / 512 vectors: the reference's 385 fixture size in 3-D, 769,
                                           # and BASELINE config 4's grid in double precision
                                           (385, capi.MG_F64, 6 / 7), (769, capi.MG_F32, 1.0), (769, capi.MG_F64, 6 / 7),
                                           (1025, capi.MG_F64, 6 / 7)])
def test_fused_double_sweep_equals_two_sweeps(n, dtype, omega):
    """k_jacobi2 (two Jacobi sweeps in one pass, row widths 64 ... 512 vectors) against two
    oracle sweeps, bit for bit, on whole grids including the 513^3 headline size."""
    kw = dict(dim=3, n=n, levels=2, dtype=dtype, length=1.0, alpha=1.0, omega=omega)
    s, ops, do = pair(**kw)
    rng = np.random.default_rng(9)
    with s:
        u = rng.random((n, n, n)).astype(s.np); b = rng.random((n, n, n)).astype(s.np)
        s.set_array(capi.ARR_U, 0, u); s.set_array(capi.ARR_RHS, 0, b)
        s.smooth(0, capi.SMOOTH_JACOBI, 2, capi.ARR_U, capi.ARR_RHS)      # one fused pair
        ref2 = ops.smooth(0, po.SMOOTH_JACOBI, 2, u, b)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ref2)
        s.smooth(0, capi.SMOOTH_JACOBI, 3, capi.ARR_U, capi.ARR_RHS)      # a pair + a single sweep
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ops.smooth(0, po.SMOOTH_JACOBI, 3, ref2, b))


@pytest.mark.parametrize("n,dtype", [(129, capi.MG_F64), (257, capi.MG_F64), (257, capi.MG_F32), (513, capi.MG_F32), (385, capi.MG_F64)])
def test_fused_red_black_sweep_equals_two_colour_passes(n, dtype):
    """k_jacobi2<RB>: a whole red-black Gauss-Seidel sweep (red half-sweep on plane p, black half-sweep
    on plane p-1 in the same pass) against the oracle's in-place colour sweeps, bit for bit."""
    kw = dict(dim=3, n=n, levels=2, dtype=dtype, length=1.0, alpha=1.0, omega=1.0)
    s, ops, do = pair(**kw)
    rng = np.random.default_rng(11)
    with s:
        u = rng.random((n, n, n)).astype(s.np); b = rng.random((n, n, n)).astype(s.np)
        s.set_array(capi.ARR_U, 0, u); s.set_array(capi.ARR_RHS, 0, b)
        s.smooth(0, capi.SMOOTH_RBGS, 1, capi.ARR_U, capi.ARR_RHS)
        ref1 = ops.smooth(0, po.SMOOTH_RBGS, 1, u, b)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ref1)
        s.smooth(0, capi.SMOOTH_RBGS, 2, capi.ARR_U, capi.ARR_RHS)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ops.smooth(0, po.SMOOTH_RBGS, 2, ref1, b))


@pytest.mark.parametrize("case", [
    dict(dim=2, n=65, dtype=capi.MG_F64, aniso=(1.0, 1.0, 1.0)),
    dict(dim=3, n=33, dtype=capi.MG_F64, aniso=(1.0, 1.0, 1.0)),
    dict(dim=3, n=65, dtype=capi.MG_F32, aniso=(1.0, 30.0, 1.0)),
    dict(dim=3, n=129, dtype=capi.MG_F64, aniso=(1.0, 100.0, 1.0)),
    dict(dim=2, n=49, dtype=capi.MG_F32, aniso=(1.0, 20.0, 1.0)),
    dict(dim=3, n=21, dtype=capi.MG_F64, aniso=(1.0, 5.0, 2.0)),
], ids=lambda c: f"{c['dim']}d-n{c['n']}-t{c['dtype']}")
@pytest.mark.parametrize("zsm", [capi.SMOOTH_ZEBRA_Y, capi.SMOOTH_ZEBRA_X], ids=["y-lines", "x-lines"])
def test_zebra_line_smoother_bit_exact(case, zsm):
    """Zebra line Gauss-Seidel along y / along x: each colour pass solves every line of that colour with the
    Thomas algorithm -- y-lines one thread per line with lanes along x, x-lines staged through LDS in 128-byte chunks so
    that the march stays coalesced; same recurrences as the oracle => same bits."""
    case = dict(case)
    if zsm == capi.SMOOTH_ZEBRA_X:
        case["aniso"] = (case["aniso"][1], case["aniso"][0], case["aniso"][2])   # the dominant coupling along the lines
    kw = dict(levels=2, length=1.0, alpha=1.0, omega=1.0, smoother=zsm, **case)
    s, ops, do = pair(**kw)
    rng = np.random.default_rng(21)
    with s:
        shape = s.level_shape(0)
        u = rnd(rng, shape, s.np); b = rnd(rng, shape, s.np)
        s.set_array(capi.ARR_U, 0, u); s.set_array(capi.ARR_RHS, 0, b)
        s.smooth(0, zsm, 1, capi.ARR_U, capi.ARR_RHS)
        ref = ops.smooth(0, zsm, 1, u, b)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ref)
        s.smooth(0, zsm, 2, capi.ARR_U, capi.ARR_RHS)
        assert np.array_equal(s.get_array(capi.ARR_U, 0), ops.smooth(0, zsm, 2, ref, b))
        # level 1 has its own factors
        sh1 = s.level_shape(1)
        u1 = rnd(rng, sh1, s.np); b1 = rnd(rng, sh1, s.np)
        s.set_array(capi.ARR_U, 1, u1); s.set_array(capi.ARR_RHS, 1, b1)
        s.smooth(1, zsm, 1, capi.ARR_U, capi.ARR_RHS)
        assert np.array_equal(s.get_array(capi.ARR_U, 1), ops.smooth(1, zsm, 1, u1, b1))


def test_zebra_smoother_needs_a_zebra_handle():
    with capi.Solver(capi.make_desc(dim=3, n=17, levels=2, smoother=capi.SMOOTH_JACOBI)) as s:
        with pytest.raises(capi.MgError):
            s.smooth(0, capi.SMOOTH_ZEBRA_Y, 1, capi.ARR_U, capi.ARR_RHS)


@pytest.mark.parametrize("dtype", [capi.MG_F64, capi.MG_F32])
@pytest.mark.parametrize("n,dim", [(129, 3), (37, 3), (65, 2)])
def test_division_by_the_diagonal_is_ieee_exact_over_the_whole_range(n, dim, dtype):
    """The kernels divide by the constant diagonal with q = a*y, r = fma(-q, cd, a), q' = fma(r, y, q)
    (mg_geom.h div_cd) inside an exponent window and with the hardware division outside it. With
    u = 0 a Jacobi sweep returns rhs / cd, so right-hand sides drawn from ALL bit patterns (normal,
    subnormal, +-0, huge, Inf, NaN) exercise both paths and the window edges; the result must have
    the bits of the oracle's plain division (NaNs at the same places)."""
    kw = dict(dim=dim, n=n, levels=2, dtype=dtype, length=1.0, alpha=1.0, omega=1.0)
    s, ops, do = pair(**kw)
    rng = np.random.default_rng(1234)
    shape = (n, n, n) if dim == 3 else (n, n)
    it = np.uint64 if dtype == capi.MG_F64 else np.uint32
    with s:
        bits = rng.integers(0, np.iinfo(it).max, size=shape, dtype=it, endpoint=True)
        b = bits.view(s.np).copy()
        # a band of exponents around the window edges and some exact zeros of both signs
        flat = b.reshape(-1)
        edge = (2.0 ** rng.integers(-1000 if dtype == capi.MG_F64 else -70, -880 if dtype == capi.MG_F64 else -50, size=flat.size // 8))
        flat[: edge.size] = (edge * rng.uniform(1, 2, edge.size)).astype(s.np)
        flat[edge.size: edge.size + 100] = 0.0
        flat[edge.size + 100: edge.size + 200] = -0.0
        u = np.zeros(shape, s.np)
        s.set_array(capi.ARR_U, 0, u); s.set_array(capi.ARR_RHS, 0, b)
        s.smooth(0, capi.SMOOTH_JACOBI, 1, capi.ARR_U, capi.ARR_RHS)
        got = s.get_array(capi.ARR_U, 0)
        with np.errstate(all="ignore"):
            ref = ops.jacobi(0, u, b)
        nan = np.isnan(ref)
        assert np.array_equal(np.isnan(got), nan)
        iv = np.int64 if dtype == capi.MG_F64 else np.int32
        assert np.array_equal(got.view(iv)[~nan], ref.view(iv)[~nan])   # bit patterns, signed zeros included


def test_manufactured_solution_second_order():
    """3-D accuracy check with no oracle in the loop: u* = sin sin sin, error O(h^2)."""
    errs = []
    for n in (33, 65):
        kw = dict(dim=3, n=n, levels=4, dtype=capi.MG_F64, length=1.0, alpha=1.0, cycle=capi.CYCLE_V,
                  smoother=capi.SMOOTH_RBGS, nu_pre=2, nu_post=2, coarse_mode=capi.COARSE_FIXED,
                  coarse_maxit=50, outer_pre_gs=0, restriction=capi.RESTRICT_FULLW)
        with capi.Solver(capi.make_desc(**kw)) as s:
            s.set_rhs(po.fill_rhs_3d(n, 1.0, 1.0, 0))
            hist, _ = s.solve(1e-10, 30)
            assert hist[-1] <= 1e-10
            errs.append(np.abs(s.get_solution() - po.exact_3d(n, 1.0)).max())
    assert 3.5 < errs[0] / errs[1] < 4.5


def _manufactured(n, aniso, alpha=1.0, length=1.0):
    """numpy only (no oracle in the loop): u* = sin(pi x) sin(pi y) sin(pi z) on the unit cube and the
    right-hand side f = -alpha (ax dxx + ay dyy + az dzz) u* = alpha (ax+ay+az) pi^2 u*, g = 0."""
    t = np.sin(np.pi * np.arange(n) / (n - 1))
    t[0] = t[-1] = 0.0
    ustar = t[:, None, None] * t[None, :, None] * t[None, None, :]
    # array axes are (z, y, x) and aniso = (ax, ay, az); u* is symmetric in them
    return ustar, alpha * sum(aniso) * (np.pi / length) ** 2 * ustar


MANUFACTURED = [
    # every extension the reference has no code for gets one oracle-independent anchor: discretisation
    # error of the converged GPU solution against the analytic one, second order in h
    dict(id="jacobi-omega-f64", smoother=capi.SMOOTH_JACOBI, omega=6 / 7, dtype=capi.MG_F64),
    dict(id="jacobi-f32", smoother=capi.SMOOTH_JACOBI, omega=6 / 7, dtype=capi.MG_F32),
    dict(id="rbgs-f32", smoother=capi.SMOOTH_RBGS, omega=1.0, dtype=capi.MG_F32),
    dict(id="semi-aniso-jacobi", smoother=capi.SMOOTH_JACOBI, omega=0.8, dtype=capi.MG_F64, aniso=(1.0, 1.0, 0.01), semi_xy=3, levels=5),
    dict(id="semi-aniso-rbgs", smoother=capi.SMOOTH_RBGS, omega=1.0, dtype=capi.MG_F64, aniso=(1.0, 1.0, 0.0625), semi_xy=2, levels=4),
    dict(id="zebra-y", smoother=capi.SMOOTH_ZEBRA_Y, omega=1.0, dtype=capi.MG_F64, aniso=(1.0, 100.0, 1.0)),
    dict(id="zebra-x", smoother=capi.SMOOTH_ZEBRA_X, omega=1.0, dtype=capi.MG_F64, aniso=(100.0, 1.0, 1.0)),
    dict(id="inject-jacobi", smoother=capi.SMOOTH_JACOBI, omega=6 / 7, dtype=capi.MG_F64, restriction=capi.RESTRICT_INJECT),
]


@pytest.mark.parametrize("case", MANUFACTURED, ids=lambda c: c["id"])
def test_manufactured_solution_anchors_every_extension(case):
    case = dict(case); case.pop("id")
    f32 = case["dtype"] == capi.MG_F32
    aniso = case.get("aniso", (1.0, 1.0, 1.0))
    levels = case.pop("levels", 4)
    errs = []
    for n in (33, 65):
        kw = dict(dim=3, n=n, levels=levels, length=1.0, alpha=1.0, cycle=capi.CYCLE_V, nu_pre=2, nu_post=2,
                  coarse_mode=capi.COARSE_FIXED, coarse_maxit=60, outer_pre_gs=0,
                  **{"restriction": capi.RESTRICT_FULLW, **case})
        ustar, f = _manufactured(n, aniso)
        with capi.Solver(capi.make_desc(**kw)) as s:
            s.set_rhs(f)
            # fp32: the relative residual bottoms out at ~1.6e-5 (n = 33) / ~6e-5 (n = 65): cd * u carries 6 / h^2
            tol = 3e-5 * (n / 33) ** 2 if f32 else 1e-10
            hist, _ = s.solve(tol, 60)
            assert hist[-1] <= tol, hist   # converged (fp32: to its round-off floor)
            errs.append(np.abs(s.get_solution().astype(np.float64) - ustar).max())
    # u* = sin sin sin: error = C h^2 (1 + O(h^2)); fp32 adds ~1e-6 of round-off to errors of 8e-4 / 2e-4
    assert (3.3 if f32 else 3.7) < errs[0] / errs[1] < (4.7 if f32 else 4.3), errs
    assert errs[1] < 4e-4   # pi^2 h^2 / 12 at h = 1/64 is 2.0e-4


def test_invalid_descriptor_is_refused():
    with pytest.raises(capi.MgError):
        capi.Solver(capi.make_desc(n=200, levels=2))  # the reference's own defaults (utilities.hpp:16,19)


EDGE = [
    dict(dim=2, n=3, levels=1), dict(dim=2, n=5, levels=2), dict(dim=2, n=17, levels=1), dict(dim=2, n=19, levels=2),
    dict(dim=3, n=3, levels=1), dict(dim=3, n=5, levels=2), dict(dim=3, n=9, levels=3), dict(dim=3, n=11, levels=2),
]


@pytest.mark.parametrize("case", EDGE, ids=lambda c: f"{c['dim']}d-n{c['n']}-L{c['levels']}")
@pytest.mark.parametrize("smoother", [capi.SMOOTH_GS_LEX, capi.SMOOTH_JACOBI])
def test_edge_grids_smallest_and_single_level(case, smoother):
    """Smallest admissible grids (3 nodes per side on the coarsest level, one interior node),
    single-level hierarchies (the cycle degenerates to the coarse solver on the finest grid,
    SURVEY appendix A) and odd multiples m*2^(L-1)+1."""
    kw = dict(alpha=1.3, length=2.0, smoother=smoother, **case)
    sg = capi.Solver(capi.make_desc(**kw)); so = po.Solver(po.make_desc(**kw))
    rng = np.random.default_rng(21)
    with sg:
        b = rng.standard_normal(sg.level_shape(0))
        sg.set_rhs(b); so.set_rhs(b)
        for _ in range(2):
            stg, sto = sg.cycle(), so.cycle()
            assert (stg.coarse_iters, stg.coarse_flag) == (sto.coarse_iters, sto.coarse_flag)
        assert np.array_equal(sg.get_solution(), so.get_solution())


def test_zero_right_hand_side_gives_nan_norms_like_the_reference():
    """Norm() = sqrt(0/0) = NaN; `NaN > tol` and `NaN <= TOL` are both false, so the reference's
    coarse loop exits at once and its outer loop runs to MaxIter. Same here."""
    kw = dict(dim=2, n=17, levels=2, smoother=capi.SMOOTH_JACOBI)
    with capi.Solver(capi.make_desc(**kw)) as s:
        s.set_rhs(np.zeros((17, 17)))
        hist, stats = s.solve(1e-11, 3)
        assert len(hist) == 4 and np.isnan(hist).all()
        assert all(st.coarse_iters == 0 and st.coarse_flag == 0 for st in stats)
        assert not s.get_solution().any()
    o = po.Solver(po.make_desc(**kw)); o.set_rhs(np.zeros((17, 17)))
    ho, so = o.solve(1e-11, 3)
    assert len(ho) == 4 and np.isnan(ho).all() and all(st.coarse_iters == 0 for st in so)
