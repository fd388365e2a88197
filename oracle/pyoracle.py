"""ctypes binding of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/gmg_oracle.h).  The product package multigrid_prj_amd
never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

MG_F64, MG_F32 = 0, 1
SMOOTH_GS_LEX, SMOOTH_JACOBI, SMOOTH_RBGS, SMOOTH_ZEBRA_Y, SMOOTH_ZEBRA_X = 0, 1, 2, 3, 4
CYCLE_SAWTOOTH, CYCLE_V = 0, 1
RESTRICT_INJECT, RESTRICT_FULLW = 0, 1
COARSE_TOL, COARSE_FIXED = 0, 1


class MgDesc(C.Structure):
    """Mirror of include/mg_desc.h::mg_desc (shared with the HIP C-ABI)."""

    _fields_ = [
        ("dim", C.c_int32), ("n", C.c_int32), ("levels", C.c_int32), ("dtype", C.c_int32),
        ("length", C.c_double), ("alpha", C.c_double),
        ("cycle", C.c_int32), ("smoother", C.c_int32),
        ("omega", C.c_double),
        ("nu_pre", C.c_int32), ("nu_post", C.c_int32),
        ("restriction", C.c_int32), ("coarse_mode", C.c_int32),
        ("coarse_maxit", C.c_int32), ("outer_pre_gs", C.c_int32),
        ("coarse_tol", C.c_double),
        ("aniso", C.c_double * 3),
        ("dist_min_n", C.c_int32), ("semi_xy", C.c_int32),
    ]


class MgCycleStats(C.Structure):
    _fields_ = [
        ("coarse_iters", C.c_int32), ("coarse_flag", C.c_int32),
        ("coarse_relres", C.c_double), ("fine_sumsq_r", C.c_double),
    ]


def make_desc(dim=2, n=17, levels=2, dtype=MG_F64, length=10.0, alpha=1.0,
              cycle=CYCLE_SAWTOOTH, smoother=SMOOTH_JACOBI, omega=1.0, nu_pre=0, nu_post=5,
              restriction=RESTRICT_INJECT, coarse_mode=COARSE_TOL, coarse_maxit=2000,
              outer_pre_gs=2, coarse_tol=1e-1, aniso=(1.0, 1.0, 1.0), dist_min_n=0, semi_xy=0) -> MgDesc:
    """Defaults == the reference program's hard-coded values (include/mg_desc.h)."""
    d = MgDesc()
    d.dim, d.n, d.levels, d.dtype = dim, n, levels, dtype
    d.length, d.alpha = length, alpha
    d.cycle, d.smoother, d.omega = cycle, smoother, omega
    d.nu_pre, d.nu_post = nu_pre, nu_post
    d.restriction, d.coarse_mode = restriction, coarse_mode
    d.coarse_maxit, d.outer_pre_gs, d.coarse_tol = coarse_maxit, outer_pre_gs, coarse_tol
    d.aniso[0], d.aniso[1], d.aniso[2] = aniso
    d.dist_min_n = dist_min_n
    d.semi_xy = semi_xy
    return d


class CoefF64(C.Structure):
    _fields_ = [("cx", C.c_double), ("cy", C.c_double), ("cz", C.c_double), ("cd", C.c_double)]


class CoefF32(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("cd", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("gmg_oracle.c", "gmg_oracle.h", "gmg_ops.inc", "gmg_cycle.inc")]
    srcs.append(os.path.join(_HERE, "..", "include", "mg_desc.h"))
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "_build/liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def _default_threads() -> int:
    """A GPU box reports every host thread but grants a 16-CPU share per GPU: more OpenMP
    threads than that only oversubscribes."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(_default_threads()))
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_validate.argtypes = [C.POINTER(MgDesc)]
        L.orc_level_n.argtypes = [C.POINTER(MgDesc), C.c_int]
        L.orc_level_nz.argtypes = [C.POINTER(MgDesc), C.c_int]
        L.orc_level_coefficients.argtypes = [C.POINTER(MgDesc), C.c_int, C.POINTER(C.c_double)]
        L.orc_fill_rhs_2d.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.orc_fill_rhs_3d.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.c_ulonglong, C.c_void_p]
        L.orc_exact_3d.argtypes = [C.c_int, C.c_double, C.c_void_p]
        for suf, coef, real in (("f64", CoefF64, C.c_double), ("f32", CoefF32, C.c_float)):
            vp = C.c_void_p
            ci = C.c_int
            getattr(L, f"orc_jacobi_{suf}").argtypes = [ci, ci, ci, coef, real, vp, vp, vp]
            getattr(L, f"orc_gs_lex_{suf}").argtypes = [ci, ci, ci, coef, vp, vp]
            getattr(L, f"orc_rbgs_{suf}").argtypes = [ci, ci, ci, coef, vp, vp]
            f = getattr(L, f"orc_residual_{suf}"); f.argtypes = [ci, ci, ci, coef, vp, vp, vp]; f.restype = C.c_double
            f = getattr(L, f"orc_sumsq_{suf}"); f.argtypes = [C.c_size_t, vp]; f.restype = C.c_double
            getattr(L, f"orc_inject_{suf}").argtypes = [ci, ci, ci, ci, vp, vp]
            getattr(L, f"orc_restrict_fw_{suf}").argtypes = [ci, ci, ci, ci, vp, vp]
            getattr(L, f"orc_prolong_overwrite_{suf}").argtypes = [ci, ci, ci, ci, vp, vp]
            getattr(L, f"orc_prolong_add_{suf}").argtypes = [ci, ci, ci, ci, vp, vp, vp]
            getattr(L, f"orc_correct_{suf}").argtypes = [C.c_size_t, vp, vp]
            getattr(L, f"orc_smooth_{suf}").argtypes = [ci, ci, ci, ci, coef, real, ci, vp, vp, vp]
            getattr(L, f"orc_coarse_solve_{suf}").argtypes = [
                ci, ci, ci, ci, coef, real, vp, vp, vp, ci, C.c_double, ci,
                C.POINTER(ci), C.POINTER(C.c_double)]
        L.orc_mg_create.argtypes = [C.POINTER(MgDesc)]; L.orc_mg_create.restype = C.c_void_p
        L.orc_mg_destroy.argtypes = [C.c_void_p]
        for name in ("orc_mg_set_rhs", "orc_mg_set_solution", "orc_mg_get_solution", "orc_mg_get_residual"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_mg_cycle.argtypes = [C.c_void_p, C.POINTER(MgCycleStats)]
        L.orc_mg_solve.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_double), C.c_int,
                                   C.POINTER(MgCycleStats)]
        L.orc_mg_smooth_fine.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_mg_residual_fine.argtypes = [C.c_void_p]; L.orc_mg_residual_fine.restype = C.c_double
        _lib = L
    return _lib


def _np_dtype(dtype):
    return np.float64 if dtype == MG_F64 else np.float32


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def level_n(desc: MgDesc, level: int) -> int:
    return lib().orc_level_n(C.byref(desc), level)


def level_nz(desc: MgDesc, level: int) -> int:
    return lib().orc_level_nz(C.byref(desc), level)


def level_shape(desc: MgDesc, level: int):
    n = level_n(desc, level)
    return (n, n) if desc.dim == 2 else (level_nz(desc, level), n, n)


def level_coef(desc: MgDesc, level: int):
    out = (C.c_double * 4)()
    lib().orc_level_coefficients(C.byref(desc), level, out)
    return tuple(out)


def coef_struct(desc: MgDesc, level: int):
    c = level_coef(desc, level)
    return (CoefF64 if desc.dtype == MG_F64 else CoefF32)(*c)


def fill_rhs_2d(n, length, test) -> np.ndarray:
    b = np.empty((n, n), np.float64)
    lib().orc_fill_rhs_2d(n, length, test, _ptr(b))
    return b


def fill_rhs_3d(n, length, alpha, kind, seed=12345) -> np.ndarray:
    b = np.empty((n, n, n), np.float64)
    lib().orc_fill_rhs_3d(n, length, alpha, kind, seed, _ptr(b))
    return b


def exact_3d(n, length) -> np.ndarray:
    u = np.empty((n, n, n), np.float64)
    lib().orc_exact_3d(n, length, _ptr(u))
    return u


class Ops:
    """Single operators on dense per-level numpy arrays (shape (n,n) or (n,n,n))."""

    def __init__(self, desc: MgDesc):
        self.d = desc
        self.suf = "f64" if desc.dtype == MG_F64 else "f32"
        self.np = _np_dtype(desc.dtype)
        self.real = C.c_double if desc.dtype == MG_F64 else C.c_float

    def _f(self, name):
        return getattr(lib(), f"orc_{name}_{self.suf}")

    def _shape(self, level):
        return level_shape(self.d, level)

    def _chk(self, a, level):
        assert a.dtype == self.np and a.flags.c_contiguous and a.shape == self._shape(level), \
            (a.dtype, a.shape, self._shape(level))

    @staticmethod
    def _nnz(a):
        """(n, nz) of a level array: n nodes per side in x/y, nz planes (1 in 2-D)"""
        return a.shape[-1], (a.shape[0] if a.ndim == 3 else 1)

    def jacobi(self, level, u, rhs, omega=None):
        self._chk(u, level); self._chk(rhs, level)
        out = np.empty_like(u)
        om = self.d.omega if omega is None else omega
        n, nz = self._nnz(u)
        self._f("jacobi")(self.d.dim, n, nz, coef_struct(self.d, level), self.real(om),
                          _ptr(u), _ptr(rhs), _ptr(out))
        return out

    def gs_lex(self, level, u, rhs):
        self._chk(u, level); out = u.copy()
        n, nz = self._nnz(u)
        self._f("gs_lex")(self.d.dim, n, nz, coef_struct(self.d, level), _ptr(out), _ptr(rhs))
        return out

    def rbgs(self, level, u, rhs):
        self._chk(u, level); out = u.copy()
        n, nz = self._nnz(u)
        self._f("rbgs")(self.d.dim, n, nz, coef_struct(self.d, level), _ptr(out), _ptr(rhs))
        return out

    def residual(self, level, u, rhs):
        self._chk(u, level); r = np.empty_like(u)
        n, nz = self._nnz(u)
        s = self._f("residual")(self.d.dim, n, nz, coef_struct(self.d, level), _ptr(u), _ptr(rhs), _ptr(r))
        return r, s

    def sumsq(self, v):
        v = np.ascontiguousarray(v, self.np)
        return self._f("sumsq")(v.size, _ptr(v))

    def _level_of(self, a):
        """level whose (n, nz) match the array (transfers need to know whether z is kept)"""
        n, nz = self._nnz(a)
        for l in range(self.d.levels):
            if level_n(self.d, l) == n and (a.ndim == 2 or level_nz(self.d, l) == nz):
                return l
        raise ValueError(f"array shape {a.shape} is not a level of this hierarchy")

    def _semi(self, fine_level):
        return int(self.d.dim == 3 and fine_level < self.d.semi_xy)

    def inject(self, fine):
        l = self._level_of(fine)
        out = np.empty(self._shape(l + 1), self.np)
        nc, nzc = self._nnz(out)
        self._f("inject")(self.d.dim, nc, nzc, self._semi(l), _ptr(fine), _ptr(out))
        return out

    def restrict_fw(self, fine):
        l = self._level_of(fine)
        out = np.empty(self._shape(l + 1), self.np)
        nc, nzc = self._nnz(out)
        self._f("restrict_fw")(self.d.dim, nc, nzc, self._semi(l), _ptr(fine), _ptr(out))
        return out

    def prolong_overwrite(self, coarse, fine_before=None):
        lc = self._level_of(coarse)
        nc, nzc = self._nnz(coarse)
        out = np.zeros(self._shape(lc - 1), self.np) if fine_before is None else fine_before.copy()
        self._f("prolong_overwrite")(self.d.dim, nc, nzc, self._semi(lc - 1), _ptr(coarse), _ptr(out))
        return out

    def prolong_add(self, coarse, fine):
        lc = self._level_of(coarse)
        nc, nzc = self._nnz(coarse)
        out = fine.copy(); scratch = np.empty_like(fine)
        self._f("prolong_add")(self.d.dim, nc, nzc, self._semi(lc - 1), _ptr(coarse), _ptr(out), _ptr(scratch))
        return out

    def correct(self, u, e):
        u2, e2 = u.copy(), e.copy()
        self._f("correct")(u2.size, _ptr(u2), _ptr(e2))
        return u2, e2

    def smooth(self, level, smoother, sweeps, u, rhs, omega=None):
        self._chk(u, level); out = u.copy(); tmp = np.empty_like(u)
        om = self.d.omega if omega is None else omega
        n, nz = self._nnz(u)
        self._f("smooth")(smoother, self.d.dim, n, nz, coef_struct(self.d, level), self.real(om),
                          sweeps, _ptr(out), _ptr(rhs), _ptr(tmp))
        return out

    def coarse_solve(self, level, smoother, e, rhs, maxit=2000, tol=1e-1, fixed=False, omega=None):
        self._chk(e, level); out = e.copy(); tmp = np.empty_like(e)
        flag = C.c_int(0); rel = C.c_double(0)
        om = self.d.omega if omega is None else omega
        n, nz = self._nnz(e)
        its = self._f("coarse_solve")(smoother, self.d.dim, n, nz, coef_struct(self.d, level),
                                      self.real(om), _ptr(out), _ptr(rhs), _ptr(tmp), maxit, tol,
                                      int(fixed), C.byref(flag), C.byref(rel))
        return out, its, flag.value, rel.value


class Solver:
    """orc_mg_*: hierarchy + cycle + outer loop (the CPU counterpart of mg_hip.h)."""

    def __init__(self, desc: MgDesc):
        self.d = desc
        self.np = _np_dtype(desc.dtype)
        self.h = lib().orc_mg_create(C.byref(desc))
        if not self.h:
            raise ValueError(f"invalid descriptor (code {lib().orc_validate(C.byref(desc))})")
        self.shape = (desc.n,) * desc.dim

    def close(self):
        if self.h:
            lib().orc_mg_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rhs(self, b):
        b = np.ascontiguousarray(b, self.np); assert b.shape == self.shape
        lib().orc_mg_set_rhs(self.h, _ptr(b))

    def set_solution(self, u):
        u = np.ascontiguousarray(u, self.np); assert u.shape == self.shape
        lib().orc_mg_set_solution(self.h, _ptr(u))

    def get_solution(self):
        u = np.empty(self.shape, self.np); lib().orc_mg_get_solution(self.h, _ptr(u)); return u

    def get_residual(self):
        r = np.empty(self.shape, self.np); lib().orc_mg_get_residual(self.h, _ptr(r)); return r

    def cycle(self) -> MgCycleStats:
        st = MgCycleStats(); lib().orc_mg_cycle(self.h, C.byref(st)); return st

    def solve(self, tol=1e-11, maxit=1000):
        hist = (C.c_double * (maxit + 1))()
        stats = (MgCycleStats * maxit)()
        nh = lib().orc_mg_solve(self.h, tol, maxit, hist, maxit + 1, stats)
        return np.array(hist[:nh]), list(stats[:nh - 1])

    def smooth_fine(self, smoother, sweeps=1):
        lib().orc_mg_smooth_fine(self.h, smoother, sweeps)

    def residual_fine(self) -> float:
        return lib().orc_mg_residual_fine(self.h)
