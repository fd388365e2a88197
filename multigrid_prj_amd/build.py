"""Compiles the HIP sources of this package for gfx950 into multigrid_prj_amd/lib/libmg_hip.so.

Plain `hipcc -shared` (no torch extension): the boundary is a C-ABI shared library.
hipcc cross-compiles without a GPU, so this also runs in the build container.
"""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmg_hip.so")

SOURCES = ["mg_kernels.hip", "mg_jacobi_fast.hip", "mg_pair_wide.hip", "mg_rr_wide.hip", "mg_transfer_fast.hip", "mg_small_levels.hip", "mg_solver.cpp", "mg_dist.cpp", "mg_capi.cpp"]
HEADERS = ["mg_geom.h", "mg_kernels.h", "mg_solver.h", "mg_comm.h"]
# -ffp-contract=off: products and sums round separately, like the reference built for
# baseline x86-64 -- required for bit parity with the oracle.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def _inputs():
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    files += [os.path.join(ROOT, "include", f) for f in ("mg_hip.h", "mg_desc.h")]
    return [f for f in files if os.path.exists(f)]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _inputs())


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (only the stale ones are recompiled, in parallel), then one link."""
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in FLAGS if f != "-shared"] + ["-I" + os.path.join(ROOT, "include")]
    with_rccl = os.environ.get("MG_WITH_RCCL", "1") == "1" and os.path.exists("/opt/rocm/lib/librccl.so")
    if with_rccl:
        flags.append("-DMG_WITH_RCCL=1")
    hdr_time = max(os.path.getmtime(f) for f in _inputs() if f.endswith((".h", ".hpp")))
    srcs = [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    jobs = []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(obj_dir, f + (".rccl" if with_rccl else "") + ".o")
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
        jobs.append((src, obj, stale))

    # several processes may find the library stale at once (the ranks of bench.py --gpus N, tests/dist_worker.py): temporary
    # names are per process and the whole build runs under a file lock, so nobody links a half-written object
    tag = f".tmp{os.getpid()}"

    def compile_one(job):
        src, obj, stale = job
        if stale:
            cmd = [hipcc, *flags, "-c", src, "-o", obj + tag]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(obj + tag, obj)
        return obj

    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():     # somebody else built it while we waited
            return LIB_PATH
        jobs = [(src, obj, force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time))
                for src, obj, _ in jobs]
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            objs = list(ex.map(compile_one, jobs))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH + tag]
        if with_rccl:
            cmd += ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(LIB_PATH + tag, LIB_PATH)
    return LIB_PATH


CLI_PATH = os.path.join(LIB_DIR, "Multigrid")
HOST = os.path.join(_HERE, "host")


def build_cli(force: bool = False) -> str:
    """The `Multigrid` executable (host C++ only; g++), linked against the in-tree library.
    The CMake target of the same name (CMakeLists.txt) builds the same sources."""
    build()
    srcs = [os.path.join(HOST, "main.cpp"), os.path.join(HOST, "utilities.cpp")]
    deps = srcs + [os.path.join(HOST, "utilities.hpp"), os.path.join(ROOT, "include", "multigrid_hip.hpp"), LIB_PATH]
    if force or not os.path.exists(CLI_PATH) or any(os.path.getmtime(f) > os.path.getmtime(CLI_PATH) for f in deps):
        subprocess.run(["g++", "-std=c++20", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, *srcs,
                        "-L" + LIB_DIR, "-lmg_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib",
                        "-o", CLI_PATH], check=True)
    return CLI_PATH


def build_mirror_harness(out_path: str, defines=()) -> str:
    """Compiles oracle/ref_harness.cpp -- written against the REFERENCE classes -- unchanged
    against include/multigrid_hip.hpp (tests/cpp/shim/allIncludes.hpp): the drop-in check."""
    build()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    subprocess.run(["g++", "-std=c++20", "-O2", "-w", *[f"-D{d}" for d in defines],
                    "-I" + os.path.join(ROOT, "tests", "cpp", "shim"),
                    "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
                    os.path.join(ROOT, "oracle", "ref_harness.cpp"), os.path.join(HOST, "utilities.cpp"),
                    "-L" + LIB_DIR, "-lmg_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", out_path], check=True)
    return out_path


REFERENCE_MAIN = "/root/reference/GeometricMultigrid/src/main.cpp"
REFMAIN_PATH = os.path.join(ROOT, "tests", "cpp", "_build", "refmain_mirror")


def build_reference_main(out_path: str = REFMAIN_PATH):
    """Compiles AND LINKS the reference's own src/main.cpp -- where it lies under /root/reference,
    nothing is copied -- against include/multigrid_hip.hpp + host/utilities.cpp + libmg_hip.so
    (INTEGRATION.md §B). Returns None where the reference is not mounted (the GPU box): the binary
    built in the container travels there with the snapshot."""
    if not os.path.exists(REFERENCE_MAIN):
        return out_path if os.path.exists(out_path) else None
    build()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    subprocess.run(["g++", "-std=c++20", "-O2", "-Wall", "-Wno-unused-parameter",
                    "-I" + os.path.join(ROOT, "tests", "cpp", "shim"),
                    "-I" + os.path.join(ROOT, "include"), "-I" + HOST,
                    REFERENCE_MAIN, os.path.join(HOST, "utilities.cpp"),
                    "-L" + LIB_DIR, "-lmg_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", out_path], check=True)
    return out_path


if __name__ == "__main__":
    print(build(force=True, verbose=True))
