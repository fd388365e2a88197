"""The drop-in boundary: the `Multigrid` executable (CLI, stdout protocol, MGGS4.txt / x.mtx)
and the C++ mirror of the reference classes (include/multigrid_hip.hpp).

The mirror is exercised by compiling oracle/ref_harness.cpp -- a driver written against the
REFERENCE's classes -- unchanged against our header, and comparing what it produces on the
GPU with the golden vectors the same driver produced from the real reference."""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from multigrid_prj_amd import build as mgbuild

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
G = os.path.join(ROOT, "tests", "golden")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "Multigrid_ref")


def run_cli(exe, args, cwd):
    p = subprocess.run([exe, *args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    return p.returncode, p.stdout


ERROR_CASES = [
    (["--help"], "Usage: ./Multigrid [OPTIONS]"),
    (["-n", "abc"], "Error: Please, insert a number after -n"),
    (["-n", "0"], "Error: Please, insert a valid N value"),
    (["-n"], "Error: Please, insert something"),
    (["-n", "33", "-ml", "0"], "Error: Please, insert a valid level"),
    (["-n", "33", "-test", "-1"], "Error: Please, insert a valid test number"),
    (["-n", "33", "-w", "-2"], "Error: Please, insert a valid width"),
    (["-n", "33", "-ml", "x"], "Error: Please, insert a number after -ml"),
]


@pytest.mark.parametrize("args,expect", ERROR_CASES, ids=[" ".join(a) for a, _ in ERROR_CASES])
def test_cli_errors_go_to_stdout_and_exit_1(args, expect, tmp_path):
    """`Error: …` on stdout + exit(1) is what WebInterface/home.php:106-113 keys on."""
    exe = mgbuild.build_cli()
    rc, out = run_cli(exe, args, tmp_path)
    assert rc == 1 and expect in out
    if os.path.exists(REF_BIN):  # the real reference, compiled in the build container
        rc_ref, out_ref = run_cli(REF_BIN, args, tmp_path)
        assert rc_ref == rc
        ours = [l for l in out.splitlines() if "MI355X extensions" not in l and not l.startswith("  -dim")]
        assert ours == out_ref.splitlines()


def test_cli_echo_lines_match_reference(tmp_path):
    exe = mgbuild.build_cli()
    args = ["-n", "33", "-a", "2.5", "-w", "4", "-ml", "3", "-test", "7", "-smt", "9"]
    rc, out = run_cli(exe, args, tmp_path)

    def echo_lines(text):  # everything before the run starts (or, without a GPU, fails to)
        lines = []
        for line in text.splitlines():
            if line.startswith(("Initialization time", "Error:", "Openmp enabled")):
                break
            lines.append(line)
        return lines

    head = echo_lines(out)
    assert head == ["Inserted N = 33", "Inserted alpha = 2.5", "Inserted width = 4", "Inserted level = 3",
                    "Inserted test number = 7", "Inserted Smoother number = 9",
                    "Warning: Invalid test case index. Default test case selected."]
    if os.path.exists(REF_BIN):
        _, out_ref = run_cli(REF_BIN, args, tmp_path)
        assert echo_lines(out_ref) == head
    rc, out = run_cli(exe, [], tmp_path)
    assert out.splitlines()[:6] == ["Inserted by default N = 200", "Inserted by default alpha = 10",
                                    "Inserted by default width = 10", "Inserted by default multigrid level = 2",
                                    "Inserted by default test number 1", "Inserted by default Smooter number 0"]
    # the reference's defaults (n=200, 2 levels) read out of range there; here they are refused
    assert rc == 1 and "Error:" in out


def test_cmake_target_multigrid_configures_and_builds(tmp_path):
    """The reference's CMakeLists.txt does not configure; ours must: target
    `Multigrid` in build/bin, option BUILD_PARALLEL."""
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    b = tmp_path / "build"
    subprocess.run(["cmake", "-S", ROOT, "-B", str(b), "-DBUILD_PARALLEL=ON"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["cmake", "--build", str(b), "-j", "4"], check=True, stdout=subprocess.DEVNULL)
    assert (b / "bin" / "Multigrid").exists() and (b / "lib" / "libmg_hip.so").exists()


def test_reference_main_cpp_builds_against_the_mirror(tmp_path):
    """SURVEY §8b-2 / INTEGRATION.md B: the reference's OWN src/main.cpp, compiled where it lies, builds
    and links against include/multigrid_hip.hpp + host/utilities.cpp + libmg_hip.so -- including
    Utils::Initialization_for_N (main.cpp:16) and the never-applied BiCGSTAB cycle (main.cpp:56).
    Container only: the reference does not exist on the GPU box."""
    if not os.path.exists(mgbuild.REFERENCE_MAIN):
        pytest.skip("/root/reference is not mounted here")
    exe = mgbuild.build_reference_main(str(tmp_path / "refmain"))
    assert exe and os.path.exists(exe)
    rc, out = run_cli(exe, ["--help"], tmp_path)   # host-only path of the binary: usage + exit(1)
    assert rc == 1 and "Usage: ./Multigrid [OPTIONS]" in out
    rc, out = run_cli(exe, ["-n", "abc"], tmp_path)
    assert rc == 1 and "Error: Please, insert a number after -n" in out


def test_cli_rejects_a_negative_maxit(tmp_path):
    exe = mgbuild.build_cli()
    rc, out = run_cli(exe, ["-n", "33", "-ml", "3", "-maxit", "-5"], tmp_path)
    assert rc == 1 and "Error: Please, insert a valid -maxit value" in out


def test_mirror_matrix_accessors_match_the_reference_semantics(tmp_path):
    """PoissonMatrix::nonZerosInRow / Domain::inRowConnections / Utils::saveMatrixOnFile
    (linear_system.hpp:44-46, domain.cpp:26-34, utilities.hpp:27-41): host-only, no GPU needed."""
    src = tmp_path / "m.cpp"
    src.write_text('''#include "allIncludes.hpp"
int main() {
    MultiGrid::SquareDomain d(5, 4.0, 0), d1(5, 4.0, 1);
    MultiGrid::PoissonMatrix<double> A(d, 2.0), A1(d1, 2.0);
    if (A.nonZerosInRow(0).size() != 1 || A.nonZerosInRow(0)[0] != 0) return 1;          // Dirichlet row: {l}
    const std::vector<size_t> r = A.nonZerosInRow(6);                                        // interior (1,1)
    if (r != std::vector<size_t>{1, 5, 6, 7, 11}) return 2;
    if (A1.nonZerosInRow(4) != std::vector<size_t>{1, 3, 4, 5, 7}) return 3;                 // level 1: width 3, centre node
    if (A.coeffRef(6, 6) != 4. * 2.0 / 1.0 || A.coeffRef(6, 7) != -2.0 / 1.0 || A.coeffRef(0, 0) != 1.) return 4;
    if (A1.coeffRef(4, 4) != 4. * 2.0 / 4.0 || A1.coeffRef(4, 1) != -2.0 / 4.0) return 5;   // h doubles per level
    Utils::saveMatrixOnFile(A, "A.txt");
    return 0;
}
''')
    exe = tmp_path / "m"
    subprocess.run(["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "tests", "cpp", "shim"),
                    "-I" + os.path.join(ROOT, "include"), "-I" + mgbuild.HOST, str(src),
                    os.path.join(mgbuild.HOST, "utilities.cpp"), "-L" + mgbuild.LIB_DIR, "-lmg_hip",
                    "-Wl,-rpath," + mgbuild.LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    mgbuild.build()
    p = subprocess.run([str(exe)], cwd=tmp_path)
    assert p.returncode == 0
    lines = open(tmp_path / "A.txt").read().split("\n")
    assert lines[0] == "25 25 61"            # rows cols nonZeros (25 + 4 * 9 interior connections)
    assert lines[1] == "0 0 1" and "6 6 8" in lines and "6 7 -2" in lines
    assert len([l for l in lines if l]) == 1 + 16 + 9 * 5   # 16 boundary rows of 1 entry, 9 interior rows of 5


@pytest.mark.gpu
@pytest.mark.parametrize("args,key", [("-n 65 -a 2.5 -w 4 -ml 4 -test 1 -smt 1", "n65_a2.5_w4_ml4_t1_s1"),
                                      ("-n 33 -a 1 -w 10 -ml 3 -test 1 -smt 0", "n33_a1_w10_ml3_t1_s0"),
                                      ("-n 257 -a 1 -w 10 -ml 3 -test 1 -smt 1", "n257_a1_w10_ml3_t1_s1"),
                                      ("-n 385 -a 1 -w 10 -ml 5 -test 0 -smt 2", "n385_a1_w10_ml5_t0_s2")])
def test_reference_main_cpp_on_the_mirror_reproduces_the_reference_runs(args, key, tmp_path):
    """The reference's unmodified main() (built in the container by __graft_entry__.build() from the
    file where it lies; the binary travels, the source does not) driving the HIP kernels through the
    mirror classes: same stdout protocol and the reference's own history -- incl. BASELINE config 1 and
    the `-smt 2` (BiCGSTAB -> Jacobi cycle) fixture run."""
    exe = mgbuild.build_reference_main()
    if not exe:
        pytest.skip("tests/cpp/_build/refmain_mirror was not built (needs /root/reference at build time)")
    rc, out = run_cli(exe, args.split(), tmp_path)
    assert rc == 0, out
    case = [c for c in json.load(open(os.path.join(G, "ref_solve.json"))) if c["key"] == key][0]
    ref = np.array([float(x) for x in case["hist"]])
    hist = [float(x) for x in open(tmp_path / "MGGS4.txt").read().split()]
    assert abs(int(hist[0]) - len(ref)) <= 1
    m = min(len(hist) - 1, len(ref))
    np.testing.assert_allclose(hist[1:1 + m], ref[:m], rtol=2e-3)
    np.testing.assert_allclose(hist[1:5], ref[:4], rtol=1e-5)     # the file holds 6 significant digits
    assert ("BiCGSTAB iters" if key.endswith("s2") else ("Jacobi iters" if key.endswith("s1") else "GS iters")) in out
    assert "||Solving elapsed time: " in out and "Tol: 1e-11<br>" in out and "Max iter: 1000<br>" in out
    assert out.count("Achieved residual on coarse grid: ") == int(hist[0]) - 1


@pytest.mark.gpu
@pytest.mark.parametrize("fix,args", [("web", "-n 145 -a 1 -w 10 -ml 5 -test 1 -smt 1"),
                                      ("gmgtest", "-n 385 -a 1 -w 10 -ml 5 -test 0 -smt 2")])
def test_cli_reproduces_the_reference_result_files(fix, args, tmp_path):
    exe = mgbuild.build_cli()
    rc, out = run_cli(exe, args.split(), tmp_path)
    assert rc == 0, out
    tail = out.split("||")[1]  # home.php:120 prints everything after `||`
    assert tail.startswith("Solving elapsed time: ") and "sec<br>" in tail
    assert "Tol: 1e-11<br>" in tail and "Max iter: 1000<br>" in tail
    hist = [float(x) for x in open(tmp_path / "MGGS4.txt").read().split()]
    ref = [float(x) for x in open(os.path.join(G, f"fixture_{fix}_MGGS4.txt")).read().split()]
    assert hist[0] == ref[0] == len(ref) - 1
    np.testing.assert_allclose(hist[1:], ref[1:], rtol=2e-3)
    assert out.count("Achieved residual on coarse grid: ") == len(ref) - 2
    x = np.loadtxt(tmp_path / "x.mtx")
    xr = np.load(os.path.join(G, f"fixture_{fix}_x.npz"))["x"]
    assert int(x[0]) == xr.size
    np.testing.assert_allclose(x[1:], xr, rtol=2e-5, atol=1e-8)


@pytest.mark.gpu
def test_cli_extension_flags_3d_vcycle(tmp_path):
    """-dim 3 -cycle v … : the MI355X extensions of the command line, against the oracle."""
    from oracle import pyoracle as po
    exe = mgbuild.build_cli()
    n = 33
    rc, out = run_cli(exe, f"-n {n} -a 1 -w 1 -ml 3 -test 0 -smt 1 -dim 3 -cycle v -omega 0.857142857142857 "
                           f"-nu1 2 -nu2 2 -fw -coarse_fixed 30 -maxit 6".split(), tmp_path)
    assert rc == 0, out
    hist = [float(x) for x in open(tmp_path / "MGGS4.txt").read().split()][1:]
    d = po.make_desc(dim=3, n=n, levels=3, length=1.0, alpha=1.0, cycle=po.CYCLE_V, smoother=po.SMOOTH_JACOBI,
                     omega=0.857142857142857, nu_pre=2, nu_post=2, restriction=po.RESTRICT_FULLW,
                     coarse_mode=po.COARSE_FIXED, coarse_maxit=30, outer_pre_gs=0)
    o = po.Solver(d)
    b = np.ones((n, n, n)); b[0] = b[-1] = 0; b[:, 0] = b[:, -1] = 0; b[:, :, 0] = b[:, :, -1] = 0  # f = 1, g = 0
    o.set_rhs(b)
    href, _ = o.solve(1e-11, 6)
    np.testing.assert_allclose(hist, href, rtol=2e-5)   # file has 6 significant digits
    x = np.loadtxt(tmp_path / "x.mtx")
    assert int(x[0]) == n ** 3
    np.testing.assert_allclose(x[1:], o.get_solution().ravel(), rtol=2e-5, atol=1e-9)


@pytest.fixture(scope="module")
def mirror_ops():
    return mgbuild.build_mirror_harness(os.path.join(ROOT, "tests", "cpp", "_build", "mirror_ops"))


def _run_op(exe, op, n, levels, level, alpha, length, smt, test, u=None, b=None):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        if u is not None:
            np.concatenate([u.ravel(), b.ravel()]).astype("<f8").tofile(fin)
        else:
            open(fin, "wb").close()
        subprocess.run([exe, op, str(n), str(levels), str(level), repr(alpha), repr(length), str(smt), str(test),
                        fin, fout], check=True, timeout=600)
        return np.fromfile(fout, "<f8")


@pytest.mark.gpu
def test_mirror_classes_reproduce_the_reference_operator_by_operator(mirror_ops):
    """Jacobi_iteration, Gauss_Seidel_iteration, Residual, InterpolationClass, Solver and
    SawtoothMGIteration of include/multigrid_hip.hpp, driven by the reference-oriented
    harness, against golden outputs of the real reference classes: bit for bit."""
    z = np.load(os.path.join(G, "ref_ops.npz"))
    for m in json.loads(bytes(z["meta_json"]).decode()):
        key, n, L, alpha, length = m["key"], m["n"], m["levels"], m["alpha"], m["length"]
        u, b = z[f"{key}_u"], z[f"{key}_b"]
        for l in range(L):
            st = 2 ** l
            o = _run_op(mirror_ops, "jacobi", n, L, l, alpha, length, 1, 0, u, b).reshape(n, n)
            assert np.array_equal(o[::st, ::st], z[f"{key}_jacobi_l{l}"]), (key, l)
            o = _run_op(mirror_ops, "gs", n, L, l, alpha, length, 0, 0, u, b).reshape(n, n)
            assert np.array_equal(o[::st, ::st], z[f"{key}_gs_l{l}"]), (key, l)
            o = _run_op(mirror_ops, "residual", n, L, l, alpha, length, 0, 0, u, b)
            assert np.array_equal(o[:n * n].reshape(n, n)[::st, ::st], z[f"{key}_residual_l{l}"]), (key, l)
            assert o[n * n] == pytest.approx(float(z[f"{key}_residual_norm_l{l}"]), rel=1e-12)
            if l < L - 1:
                o = _run_op(mirror_ops, "interp", n, L, l, alpha, length, 0, 0, u, b).reshape(n, n)
                assert np.array_equal(o[::st, ::st], z[f"{key}_interp_l{l}"]), (key, l)
        lc, st = L - 1, 2 ** (L - 1)
        for smt in (0, 1):
            # the harness wraps the smoother in its own CountingSmoother subclass, so this goes
            # through Solver's generic operator-by-operator loop
            o = _run_op(mirror_ops, "coarse_solve", n, L, lc, alpha, length, smt, 0, np.zeros_like(u), b)
            norm, sweeps, status = z[f"{key}_coarse_smt{smt}_stats"]
            assert (o[n * n + 1], o[n * n + 2]) == (sweeps, status)
            assert np.array_equal(o[:n * n].reshape(n, n)[::st, ::st], z[f"{key}_coarse_smt{smt}_e"])
            o = _run_op(mirror_ops, "cycle", n, L, 0, alpha, length, smt, 0, u, b).reshape(n, n)
            assert np.array_equal(o, z[f"{key}_cycle_smt{smt}"]), (key, smt)


@pytest.mark.gpu
def test_mirror_main_loop_reproduces_the_reference_history(mirror_ops):
    """`u * GS * GS * MG1; u * RES` exactly as the reference's main() writes it."""
    cases = [c for c in json.load(open(os.path.join(G, "ref_solve.json"))) if c["n"] <= 65]
    for c in cases:
        smt = c["smt"]
        o = _run_op(mirror_ops, "solve_full", c["n"], c["levels"], 0, float(c["alpha"]), float(c["length"]), smt, c["test"])
        nh = int(o[0])
        ref = np.array([float(x) for x in c["hist"]])
        assert abs(nh - len(ref)) <= 1
        m = min(nh, len(ref))
        np.testing.assert_allclose(o[1:1 + m], ref[:m], rtol=2e-3)
        np.testing.assert_allclose(o[1:5], ref[:4], rtol=1e-9)


def _expected_frames(po, d, u, b):
    """The frames the reference's CREATE_GIF twin saves for one cycle (multigrid.hpp:212-299),
    rebuilt from oracle operators: sol (+ err) sampled on the level being worked on."""
    ops = po.Ops(d)
    L = d.levels
    r, _ = ops.residual(0, u, b)
    rhs = [r]
    for l in range(1, L):
        rhs.append(ops.inject(rhs[-1]))
    samp = lambda l: np.ascontiguousarray(u[::2 ** l, ::2 ** l])
    frames = [samp(L - 1)]
    e = ops.coarse_solve(L - 1, d.smoother, np.zeros_like(rhs[L - 1]), rhs[L - 1], maxit=d.coarse_maxit, tol=d.coarse_tol)[0]
    frames.append(samp(L - 1) + e)
    for l in range(L - 2, -1, -1):
        e = ops.prolong_overwrite(e)
        frames.append(samp(l) + e)
        e = ops.smooth(l, d.smoother, d.nu_post, e, rhs[l])
        frames.append(samp(l) + e)
    frames.append(u + e)
    return frames


@pytest.mark.gpu
def test_stage_callback_reproduces_create_gif_frames():
    """mg_set_stage_callback == the reference's -DCREATE_GIF stage dumps, bit for bit."""
    from multigrid_prj_amd import capi
    from oracle import pyoracle as po
    kw = dict(dim=2, n=33, levels=3, alpha=1.0, length=10.0, smoother=capi.SMOOTH_JACOBI, nu_post=2, coarse_tol=0.6)
    rng = np.random.default_rng(11)
    u, b = rng.standard_normal((33, 33)), rng.standard_normal((33, 33))
    got = []
    with capi.Solver(capi.make_desc(**kw)) as s:
        s.set_rhs(b); s.set_solution(u)
        s.set_stage_callback(lambda stage, level, a: got.append((stage, level, a)))
        s.cycle()
        s.set_stage_callback(None)
        s.cycle()  # no more frames
    exp = _expected_frames(po, po.make_desc(**kw), u, b)
    assert [g[0] for g in got] == list(range(len(exp))) == list(range(2 + 2 * 2 + 1))
    assert [g[1] for g in got] == [2, 2, 1, 1, 0, 0, 0]
    for (stage, level, a), e in zip(got, exp):
        assert np.array_equal(a, e), stage


@pytest.mark.gpu
def test_mirror_with_create_gif_writes_the_reference_frame_files(tmp_path):
    """Built with -DCREATE_GIF the mirror's SawtoothMGIteration switches to nu = 2 / tolerance 0.6
    and writes ./output/<k>.mtx after every stage, like the reference twin gifMaker.py reads."""
    exe = mgbuild.build_mirror_harness(os.path.join(ROOT, "tests", "cpp", "_build", "mirror_ops_gif"), defines=("CREATE_GIF",))
    z = np.load(os.path.join(G, "ref_ops.npz"))
    m = json.loads(bytes(z["meta_json"]).decode())[0]
    n, L = m["n"], m["levels"]
    u, b = z[f"{m['key']}_u"], z[f"{m['key']}_b"]
    os.makedirs(tmp_path / "output")
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    np.concatenate([u.ravel(), b.ravel()]).astype("<f8").tofile(fin)
    subprocess.run([exe, "cycle", str(n), str(L), "0", repr(m["alpha"]), repr(m["length"]), "1", "0", str(fin), str(fout)],
                   check=True, cwd=tmp_path, timeout=300)
    frames = sorted(os.listdir(tmp_path / "output"), key=lambda f: int(f.split(".")[0]))
    assert frames == [f"{k}.mtx" for k in range(2 + 2 * (L - 1) + 1)]
    last = np.loadtxt(tmp_path / "output" / frames[-1])
    assert int(last[0]) == n * n
    np.testing.assert_allclose(last[1:], np.fromfile(fout, "<f8"), rtol=2e-5, atol=1e-9)
    first = np.loadtxt(tmp_path / "output" / "0.mtx")
    nc = (n - 1) // 2 ** (L - 1) + 1
    assert int(first[0]) == nc * nc
    np.testing.assert_allclose(first[1:], u[::2 ** (L - 1), ::2 ** (L - 1)].ravel(), rtol=2e-5, atol=1e-9)
