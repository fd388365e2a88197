"""ctypes binding of libmg_hip.so (include/mg_hip.h) -- the host-side mirror used by
tests and bench.py.  There is no CPU fallback: loading fails loudly when the HIP
library is missing, and Solver() fails loudly when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

MG_F64, MG_F32 = 0, 1
SMOOTH_GS_LEX, SMOOTH_JACOBI, SMOOTH_RBGS, SMOOTH_ZEBRA_Y, SMOOTH_ZEBRA_X = 0, 1, 2, 3, 4
CYCLE_SAWTOOTH, CYCLE_V = 0, 1
RESTRICT_INJECT, RESTRICT_FULLW = 0, 1
COARSE_TOL, COARSE_FIXED = 0, 1
ARR_U, ARR_E, ARR_RHS, ARR_TMP, ARR_RES = 0, 1, 2, 3, 4
MG_COMM_ID_BYTES = 128
PROF_SMOOTH, PROF_SMOOTH_PROLONG, PROF_RESID_RESTRICT, PROF_PROLONG = 0, 1, 2, 3


class MgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmg_hip error {code}: {msg}")
        self.code = code


class MgDesc(C.Structure):
    """include/mg_desc.h::mg_desc"""

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
    """Defaults are the reference program's hard-coded values (include/mg_desc.h)."""
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


class MgP2POp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("is_send", C.c_int32), ("buf", C.c_void_p), ("bytes", C.c_size_t)]


STAGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(MgP2POp), C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class MgHostComm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("batch", BATCH_FN), ("allreduce_sum", ALLREDUCE_FN)]


# every symbol include/mg_hip.h declares (tests check the library exports them all)
EXPORTS = [
    "mg_last_error", "mg_device_count", "mg_create", "mg_destroy", "mg_level_n", "mg_level_nz",
    "mg_level_coefficients", "mg_set_rhs", "mg_set_solution", "mg_get_solution", "mg_set_array",
    "mg_get_array", "mg_zero_array", "mg_smooth", "mg_residual", "mg_sumsq", "mg_restrict",
    "mg_prolong", "mg_correct", "mg_coarse_solve", "mg_coarse_solve_ex", "mg_cycle", "mg_cycle_async", "mg_solve", "mg_solve_lockstep",
    "mg_set_stage_callback", "mg_sync", "mg_timer_start", "mg_timer_stop", "mg_profile_begin", "mg_profile_end", "mg_profile_fused", "mg_profile_get", "mg_comm_info", "mg_comm_stats", "mg_device_bytes", "mg_comm_unique_id", "mg_comm_selftest",
    "mg_create_distributed", "mg_create_distributed_hostcomm", "mg_create_distributed_dryrun", "mg_plan_slab",
]

_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """Loads (building first if stale) the in-tree libmg_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if build_if_missing:
        path = _build.build()
    if not os.path.exists(path):
        raise MgError(-2, f"{path} is missing: build it with `python -m multigrid_prj_amd.build`")
    L = C.CDLL(path)
    vp, i, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    L.mg_last_error.restype = C.c_char_p
    L.mg_device_count.argtypes = [C.POINTER(i)]
    L.mg_create.argtypes = [C.POINTER(MgDesc), i, C.POINTER(vp)]
    L.mg_destroy.argtypes = [vp]
    L.mg_level_n.argtypes = [vp, i, C.POINTER(i)]
    L.mg_level_nz.argtypes = [vp, i, C.POINTER(i)]
    L.mg_level_coefficients.argtypes = [vp, i, dp]
    L.mg_set_rhs.argtypes = [vp, vp]
    L.mg_set_solution.argtypes = [vp, vp]
    L.mg_get_solution.argtypes = [vp, vp]
    L.mg_set_array.argtypes = [vp, i, i, vp]
    L.mg_get_array.argtypes = [vp, i, i, vp]
    L.mg_zero_array.argtypes = [vp, i, i]
    L.mg_smooth.argtypes = [vp, i, i, i, i, i]
    L.mg_residual.argtypes = [vp, i, i, i, i, dp]
    L.mg_sumsq.argtypes = [vp, i, i, dp]
    L.mg_restrict.argtypes = [vp, i, i, i, i]
    L.mg_prolong.argtypes = [vp, i, i, i, i]
    L.mg_correct.argtypes = [vp, i, i]
    L.mg_coarse_solve.argtypes = [vp, i, i, i, C.POINTER(MgCycleStats)]
    L.mg_coarse_solve_ex.argtypes = [vp, i, i, i, i, i, C.c_double, i, C.POINTER(MgCycleStats)]
    L.mg_cycle.argtypes = [vp, C.POINTER(MgCycleStats)]
    L.mg_cycle_async.argtypes = [vp, i]
    L.mg_solve.argtypes = [vp, C.c_double, i, dp, i, C.POINTER(i), C.POINTER(MgCycleStats)]
    L.mg_solve_lockstep.argtypes = [vp, C.c_double, i, C.POINTER(i), i, dp, i, C.POINTER(i), C.POINTER(MgCycleStats)]
    L.mg_set_stage_callback.argtypes = [vp, STAGE_FN, vp]
    L.mg_sync.argtypes = [vp]
    L.mg_timer_start.argtypes = [vp]
    L.mg_timer_stop.argtypes = [vp, dp]
    L.mg_profile_begin.argtypes = [vp]
    L.mg_profile_end.argtypes = [vp, dp, C.POINTER(i)]
    L.mg_profile_fused.argtypes = [vp, dp, C.POINTER(i)]
    L.mg_profile_get.argtypes = [vp, i, dp, C.POINTER(i)]
    L.mg_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(C.c_char_p)]
    L.mg_comm_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.mg_device_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.mg_comm_unique_id.argtypes = [vp]
    L.mg_comm_selftest.argtypes = [C.c_size_t]
    L.mg_create_distributed.argtypes = [C.POINTER(MgDesc), i, i, i, vp, C.POINTER(vp)]
    L.mg_create_distributed_hostcomm.argtypes = [C.POINTER(MgDesc), i, i, i, C.POINTER(MgHostComm), C.POINTER(vp)]
    L.mg_create_distributed_dryrun.argtypes = [C.POINTER(MgDesc), i, i, i, C.POINTER(vp)]
    L.mg_plan_slab.argtypes = [C.POINTER(MgDesc), i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise MgError(rc, load().mg_last_error().decode())


def device_count() -> int:
    n = C.c_int(0)
    _check(load().mg_device_count(C.byref(n)))
    return n.value


def plan_slab(desc: MgDesc, nranks: int, rank: int, level: int):
    """Host-only: (z0, nz, first_gathered_level) of `rank` on `level`."""
    z0, nz, fg = C.c_int(0), C.c_int(0), C.c_int(0)
    _check(load().mg_plan_slab(C.byref(desc), nranks, rank, level, C.byref(z0), C.byref(nz), C.byref(fg)))
    return z0.value, nz.value, fg.value


def comm_selftest(nbytes: int = 1 << 20):
    """RCCL transport smoke test on the current device (one rank, send/recv to self)."""
    _check(load().mg_comm_selftest(nbytes))


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(MG_COMM_ID_BYTES)
    _check(load().mg_comm_unique_id(buf))
    return buf.raw


class Solver:
    """One GPU-resident hierarchy. Mirrors the operator vocabulary of the reference
    (`x * smoother`, `x * RES`, interpolate, Solve, SawtoothMGIteration, main loop)."""

    def __init__(self, desc: MgDesc, device: int = -1, rank: int = 0, nranks: int = 1, comm_id: bytes | None = None,
                 host_comm: "MgHostComm | None" = None, dry: bool = False):
        self.lib = load()
        self.d = desc
        self.np = np.float64 if desc.dtype == MG_F64 else np.float32
        self.h = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        self._host_comm = host_comm  # keep the callbacks alive
        if nranks > 1 and dry:   # measurement only: no peers, nothing moves
            _check(self.lib.mg_create_distributed_dryrun(C.byref(desc), device, rank, nranks, C.byref(self.h)))
        elif nranks > 1 and host_comm is not None:
            _check(self.lib.mg_create_distributed_hostcomm(C.byref(desc), device, rank, nranks, C.byref(host_comm), C.byref(self.h)))
        elif nranks > 1:
            buf = C.create_string_buffer(comm_id, MG_COMM_ID_BYTES)
            _check(self.lib.mg_create_distributed(C.byref(desc), device, rank, nranks, buf, C.byref(self.h)))
        else:
            _check(self.lib.mg_create(C.byref(desc), device, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- geometry
    def level_n(self, level: int) -> int:
        n = C.c_int(0); _check(self.lib.mg_level_n(self.h, level, C.byref(n))); return n.value

    def level_nz(self, level: int) -> int:
        n = C.c_int(0); _check(self.lib.mg_level_nz(self.h, level, C.byref(n))); return n.value

    def level_shape(self, level: int):
        """Host shape of this rank's part of a level (the local z-slab when distributed)."""
        n = self.level_n(level)
        return (n, n) if self.d.dim == 2 else (self.level_nz(level), n, n)

    def level_coefficients(self, level: int):
        out = (C.c_double * 4)(); _check(self.lib.mg_level_coefficients(self.h, level, out)); return tuple(out)

    # -- data movement
    def _host(self, a, level):
        a = np.ascontiguousarray(a, self.np)
        if a.shape != self.level_shape(level):
            raise ValueError(f"expected shape {self.level_shape(level)}, got {a.shape}")
        return a

    def set_array(self, which, level, a):
        a = self._host(a, level); _check(self.lib.mg_set_array(self.h, which, level, a.ctypes.data_as(C.c_void_p)))

    def get_array(self, which, level):
        a = np.empty(self.level_shape(level), self.np)
        _check(self.lib.mg_get_array(self.h, which, level, a.ctypes.data_as(C.c_void_p))); return a

    def zero_array(self, which, level):
        _check(self.lib.mg_zero_array(self.h, which, level))

    def set_rhs(self, b): self.set_array(ARR_RHS, 0, b)
    def set_solution(self, u): self.set_array(ARR_U, 0, u)
    def get_solution(self): return self.get_array(ARR_U, 0)

    # -- operators
    def smooth(self, level, smoother, sweeps, arr_x, arr_rhs):
        _check(self.lib.mg_smooth(self.h, level, smoother, sweeps, arr_x, arr_rhs))

    def residual(self, level, arr_x, arr_rhs, arr_r=-1) -> float:
        s = C.c_double(0); _check(self.lib.mg_residual(self.h, level, arr_x, arr_rhs, arr_r, C.byref(s))); return s.value

    def residual_async(self, level, arr_x, arr_rhs, arr_r=-1):
        _check(self.lib.mg_residual(self.h, level, arr_x, arr_rhs, arr_r, None))

    def sumsq(self, level, arr) -> float:
        s = C.c_double(0); _check(self.lib.mg_sumsq(self.h, level, arr, C.byref(s))); return s.value

    def restrict(self, fine_level, kind, arr_src, arr_dst):
        _check(self.lib.mg_restrict(self.h, fine_level, kind, arr_src, arr_dst))

    def prolong(self, coarse_level, add, arr_src, arr_dst):
        _check(self.lib.mg_prolong(self.h, coarse_level, int(add), arr_src, arr_dst))

    def correct(self, arr_u=ARR_U, arr_e=ARR_E):
        _check(self.lib.mg_correct(self.h, arr_u, arr_e))

    def coarse_solve(self, level, arr_x, arr_rhs) -> MgCycleStats:
        st = MgCycleStats(); _check(self.lib.mg_coarse_solve(self.h, level, arr_x, arr_rhs, C.byref(st))); return st

    def cycle(self) -> MgCycleStats:
        st = MgCycleStats(); _check(self.lib.mg_cycle(self.h, C.byref(st))); return st

    def cycle_async(self, count=1):
        _check(self.lib.mg_cycle_async(self.h, count))

    def solve(self, tol=1e-11, maxit=1000):
        hist = (C.c_double * (maxit + 1))(); nh = C.c_int(0)
        stats = (MgCycleStats * max(maxit, 1))()
        _check(self.lib.mg_solve(self.h, tol, maxit, hist, maxit + 1, C.byref(nh), stats))
        return np.array(hist[:nh.value]), list(stats[:nh.value - 1])

    def solve_lockstep(self, coarse_counts, tol=1e-11, maxit=1000):
        """mg_solve with outer iteration i's coarse solve spending exactly coarse_counts[i] sweeps"""
        hist = (C.c_double * (maxit + 1))(); nh = C.c_int(0)
        stats = (MgCycleStats * max(maxit, 1))()
        cnt = (C.c_int * max(len(coarse_counts), 1))(*[int(c) for c in coarse_counts])
        _check(self.lib.mg_solve_lockstep(self.h, tol, maxit, cnt, len(coarse_counts), hist, maxit + 1, C.byref(nh), stats))
        return np.array(hist[:nh.value]), list(stats[:nh.value - 1])

    def set_stage_callback(self, fn):
        """fn(stage, level, array) after every stage of the sawtooth cycle (CREATE_GIF dumps); None removes it"""
        if fn is None:
            self._stage_cb = STAGE_FN(0)
        else:
            def tramp(user, stage, level, n, nz, ptr):
                shape = (n, n) if self.d.dim == 2 else (nz, n, n)
                cnt = int(np.prod(shape))
                buf = (C.c_double if self.d.dtype == MG_F64 else C.c_float) * cnt
                fn(stage, level, np.ctypeslib.as_array(buf.from_address(ptr)).reshape(shape).copy())
            self._stage_cb = STAGE_FN(tramp)
        _check(self.lib.mg_set_stage_callback(self.h, self._stage_cb, None))

    def sync(self): _check(self.lib.mg_sync(self.h))
    def timer_start(self): _check(self.lib.mg_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_double(0); _check(self.lib.mg_timer_stop(self.h, C.byref(ms))); return ms.value

    def profile_begin(self): _check(self.lib.mg_profile_begin(self.h))

    def profile_end(self):
        """-> (summed ms of the finest-grid smoother calls, number of sweeps)"""
        ms = C.c_double(0); n = C.c_int(0)
        _check(self.lib.mg_profile_end(self.h, C.byref(ms), C.byref(n))); return ms.value, n.value

    def profile_fused(self):
        """-> (summed ms, sweeps) of the finest-grid launches that also carried the prolongation
        (valid after profile_end)"""
        ms = C.c_double(0); n = C.c_int(0)
        _check(self.lib.mg_profile_fused(self.h, C.byref(ms), C.byref(n))); return ms.value, n.value

    def profile_get(self, kind):
        """-> (summed ms, launches) of one kind of finest-level launch (PROF_*; valid after profile_end)"""
        ms = C.c_double(0); n = C.c_int(0)
        _check(self.lib.mg_profile_get(self.h, kind, C.byref(ms), C.byref(n))); return ms.value, n.value

    def comm_info(self):
        """-> (rank, nranks, ranks the transport reports, transport name)"""
        r, n, t = C.c_int(0), C.c_int(0), C.c_int(0); name = C.c_char_p()
        _check(self.lib.mg_comm_info(self.h, C.byref(r), C.byref(n), C.byref(t), C.byref(name)))
        return r.value, n.value, t.value, name.value.decode()

    def comm_stats(self):
        """-> (message groups posted, bytes sent) by this rank since creation"""
        g, b = C.c_longlong(0), C.c_longlong(0)
        _check(self.lib.mg_comm_stats(self.h, C.byref(g), C.byref(b))); return g.value, b.value

    def device_bytes(self) -> int:
        b = C.c_size_t(0); _check(self.lib.mg_device_bytes(self.h, C.byref(b))); return b.value
