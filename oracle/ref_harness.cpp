// ref_harness.cpp -- TEST INFRASTRUCTURE, builds only in the container that has
// /root/reference (oracle/Makefile target `ref`, output oracle/_ref/ref_ops).
//
// Our own driver around the UNMODIFIED reference classes: it #includes the
// reference headers where they lie (-I/root/reference/GeometricMultigrid/include)
// and links the reference's own src/{domain,multigrid,utilities}.cpp. Nothing of
// the reference is copied into this repository. It applies ONE reference operator
// to vectors read from a raw little-endian double file and writes the result, so
// tests/golden/make_golden.py can pin the CPU restatement (oracle/gmg_oracle.c)
// operator by operator and cycle by cycle at full double precision.
//
// usage: ref_ops <op> <n> <levels> <level> <alpha> <length> <smt> <test> <in.bin> <out.bin>
//   in.bin : u[n*n] then b[n*n]            (ignored by op=solve_full / solve_counts)
//   out.bin: op-specific doubles, see each branch
#include "allIncludes.hpp"

#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>

using Vec = std::vector<double>;
using namespace MultiGrid;

// counts how many sweeps Solver::Solve spends (the reference keeps the counter local)
struct CountingSmoother : public SmootherClass<Vec> {
    SmootherClass<Vec> &inner;
    long count = 0;
    explicit CountingSmoother(SmootherClass<Vec> &s) : inner(s) {}
    void apply_iteration_to_vec(Vec &sol) override { inner.apply_iteration_to_vec(sol); ++count; }
};

// The same idea for a whole cycle: SawtoothMGIteration builds its smoothers itself from the template
// argument (multigrid.hpp:112-114), so a subclass that counts its applications per level width shows
// how many sweeps the cycle's coarse Solver spent (the coarsest level's smoother is applied by nobody
// else: the nu post-sweeps run on levels L-2 .. 0, multigrid.hpp:134-139).
static std::vector<long> g_applied(1 << 16, 0);
template <class Base>
struct Counted : public Base {
    size_t width;
    Counted(PoissonMatrix<double> &A, Vec &f) : Base(A, f), width(A.getWidth()) {}
    void apply_iteration_to_vec(Vec &sol) override { g_applied[width]++; Base::apply_iteration_to_vec(sol); }
};

static Vec read_doubles(FILE *f, size_t cnt)
{
    Vec v(cnt);
    if (fread(v.data(), sizeof(double), cnt, f) != cnt) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    return v;
}
static void write_doubles(FILE *f, const Vec &v) { fwrite(v.data(), sizeof(double), v.size(), f); }
static void write_double(FILE *f, double x) { fwrite(&x, sizeof(double), 1, f); }

int main(int argc, char **argv)
{
    if (argc != 11) { std::fprintf(stderr, "bad usage\n"); return 2; }
    std::string op = argv[1];
    size_t n = std::stoul(argv[2]);
    int levels = std::stoi(argv[3]);
    int level = std::stoi(argv[4]);
    double alpha = std::atof(argv[5]);
    double length = std::atof(argv[6]);
    int smt = std::stoi(argv[7]);
    int test = std::stoi(argv[8]);

    std::vector<SquareDomain> domains;
    for (int i = 0; i < levels; i++) domains.push_back(SquareDomain(n, length, i));
    std::vector<PoissonMatrix<double>> A;
    for (auto &dom : domains) A.push_back(PoissonMatrix<double>(dom, alpha));

    FILE *fout = std::fopen(argv[10], "wb");
    if (!fout) return 2;

    if (op == "solve_full") {
        // the outer loop of the reference's main(), driven at full precision
        std::function<double(const double, const double)> f, g;
        Utils::init_test_functions(f, g, test);
        DataVector<double> fvec(domains.front(), f, g);
        Vec u(n * n, 0.), res(n * n, 0.), hist, coarse;
        SawtoothMGIteration<DataVector<double>, Gauss_Seidel_iteration<Vec>> MG0(A, fvec);
        SawtoothMGIteration<DataVector<double>, Jacobi_iteration<Vec>> MG1(A, fvec);
        Residual<DataVector<double>> RES(A.front(), fvec, res);
        Gauss_Seidel_iteration<DataVector<double>> GS(A.front(), fvec);
        std::stringstream captured;
        std::streambuf *old = std::cout.rdbuf(captured.rdbuf());
        u * RES;
        hist.push_back(RES.Norm());
        for (int i = 0; i < 1000; i++) {
            if (smt == 0) u * GS * GS * MG0; else u * GS * GS * MG1;
            u * RES;
            hist.push_back(RES.Norm());
            if (hist.back() <= TOL) break;
        }
        std::cout.rdbuf(old);
        // "Achieved residual on coarse grid: <x>" lines (6 s.d.) -> coarse[]
        std::string line;
        while (std::getline(captured, line)) {
            auto p = line.find(": ");
            if (line.rfind("Achieved", 0) == 0 && p != std::string::npos)
                coarse.push_back(std::atof(line.c_str() + p + 2));
        }
        write_double(fout, (double)hist.size());
        write_doubles(fout, hist);
        write_doubles(fout, coarse);
        write_doubles(fout, u);
        Vec bvec(n * n);
        for (size_t i = 0; i < n * n; i++) bvec[i] = fvec[i];
        write_doubles(fout, bvec);
        std::fclose(fout);
        return 0;
    }

    if (op == "solve_counts") {
        // the same outer loop with counting smoothers: hist again + coarse sweeps per cycle
        std::function<double(const double, const double)> f, g;
        Utils::init_test_functions(f, g, test);
        DataVector<double> fvec(domains.front(), f, g);
        Vec u(n * n, 0.), res(n * n, 0.), hist, counts;
        SawtoothMGIteration<DataVector<double>, Counted<Gauss_Seidel_iteration<Vec>>> MG0(A, fvec);
        SawtoothMGIteration<DataVector<double>, Counted<Jacobi_iteration<Vec>>> MG1(A, fvec);
        Residual<DataVector<double>> RES(A.front(), fvec, res);
        Gauss_Seidel_iteration<DataVector<double>> GS(A.front(), fvec);
        std::stringstream captured;
        std::streambuf *old = std::cout.rdbuf(captured.rdbuf());
        const size_t wc = A.back().getWidth();
        u * RES;
        hist.push_back(RES.Norm());
        for (int i = 0; i < 1000; i++) {
            const long before = g_applied[wc];
            if (smt == 0) u * GS * GS * MG0; else u * GS * GS * MG1;
            long spent = g_applied[wc] - before;
            counts.push_back((double)spent);
            u * RES;
            hist.push_back(RES.Norm());
            if (hist.back() <= TOL) break;
        }
        std::cout.rdbuf(old);
        write_double(fout, (double)hist.size());
        write_doubles(fout, hist);
        write_doubles(fout, counts);
        write_doubles(fout, u);
        std::fclose(fout);
        return 0;
    }

    FILE *fin = std::fopen(argv[9], "rb");
    if (!fin) return 2;
    Vec u = read_doubles(fin, n * n);
    Vec b = read_doubles(fin, n * n);
    std::fclose(fin);

    if (op == "jacobi") {
        Jacobi_iteration<Vec> J(A[level], b);
        u * J;
        write_doubles(fout, u);
    } else if (op == "gs") {
        Gauss_Seidel_iteration<Vec> G(A[level], b);
        u * G;
        write_doubles(fout, u);
    } else if (op == "residual") {
        Vec res(n * n, 0.);
        Residual<Vec> R(A[level], b, res);
        u * R;
        write_doubles(fout, res);
        write_double(fout, R.Norm());
    } else if (op == "interp") {
        InterpolationClass P(A[level + 1], A[level]);
        u * P;
        write_doubles(fout, u);
    } else if (op == "coarse_solve") {
        std::unique_ptr<SmootherClass<Vec>> sm;
        if (smt == 0) sm = std::make_unique<Gauss_Seidel_iteration<Vec>>(A[level], b);
        else sm = std::make_unique<Jacobi_iteration<Vec>>(A[level], b);
        CountingSmoother cs(*sm);
        Residual<Vec> R(A[level], b);
        R.refresh_normalization_constant();
        Solver<Vec> S(cs, R, 2000, 1.e-1, 1);
        u * S * R;
        write_doubles(fout, u);
        write_double(fout, R.Norm());
        write_double(fout, (double)cs.count);
        write_double(fout, (double)S.Status());
    } else if (op == "cycle") {
        std::stringstream captured;
        std::streambuf *old = std::cout.rdbuf(captured.rdbuf());
        if (smt == 0) { SawtoothMGIteration<Vec, Gauss_Seidel_iteration<Vec>> MG(A, b); u * MG; }
        else { SawtoothMGIteration<Vec, Jacobi_iteration<Vec>> MG(A, b); u * MG; }
        std::cout.rdbuf(old);
        write_doubles(fout, u);
    } else {
        std::fprintf(stderr, "unknown op\n");
        return 2;
    }
    std::fclose(fout);
    return 0;
}
