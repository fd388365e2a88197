// utilities.hpp -- CLI, test-function table and result writers of the `Multigrid` executable.
// Same flags, defaults, messages and file formats as the reference driver
// (/root/reference/GeometricMultigrid/include/utilities.hpp, src/utilities.cpp), re-written.
#ifndef MG_HOST_UTILITIES_HPP
#define MG_HOST_UTILITIES_HPP

#include <fstream>
#include <functional>
#include <string>
#include <vector>

enum SMOOTHERS { Gauss_Siedel, Jacobi, BiCGSTAB, SMOOTHERS_END };  // utilities.hpp:9-14

#define DEFAULT_N 200  // the reference's default; refused here because 199 is odd (see Options)
#define DEFAULT_ALPHA 10.0
#define DEFAULT_WIDTH 10.0
#define DEFAULT_LEVEL 2
#define DEFAULT_TEST 1
#define DEFAULT_METHOD Gauss_Siedel

namespace Utils {

struct Options {
    // reference flags -n -a -w -ml -test -smt
    size_t N = DEFAULT_N;
    double alpha = DEFAULT_ALPHA;
    double width = DEFAULT_WIDTH;
    int level = DEFAULT_LEVEL;
    int test = DEFAULT_TEST;
    SMOOTHERS smoother = DEFAULT_METHOD;
    // extensions (absent from the reference): -dim 3, -cycle v, -omega, -nu1, -nu2, -rbgs, -zebra, -zebrax (line Gauss-Seidel along y / x), -anisox A, -anisoy A (coupling multipliers),
    // -fw, -coarse_fixed K, -fp32, -maxit, -cold (no warm-up cycle before the solve timer), -eps E (z-coupling multiplier), -semi K (first K coarsenings in x,y only)
    int dim = 2;
    bool vcycle = false, rbgs = false, zebra = false, full_weighting = false, fp32 = false, cold = false, zebrax = false;
    double omega = 1.0, eps_z = 1.0, aniso_x = 1.0, aniso_y = 1.0;
    int semi = 0;
    int nu1 = 2, nu2 = -1, coarse_fixed = -1, maxit = 1000;
};

// Parses argv like the reference's Initialization_for_N (same echo lines, same `Error: …`
// messages on stdout, exit(1) on error or --help).
void parse_command_line(int argc, char **argv, Options &opt);

// The reference's own entry point (include/utilities.hpp:25, src/utilities.cpp:3-132), so that its
// src/main.cpp builds against this directory unchanged: the six reference flags through
// parse_command_line (the extension flags are parsed too, but have nowhere to go in this signature).
void Initialization_for_N(int argc, char **argv, size_t &N, double &alpha, double &width, int &level,
                          int &functions_to_test, SMOOTHERS &sm);

// (f, g) pairs of the reference table; an index outside 0..2 selects pair 0 with a warning.
void init_test_functions(std::function<double(const double, const double)> &f,
                         std::function<double(const double, const double)> &g, int i);

// include/utilities.hpp:27-41: "rows cols nonZeros", then one "i j a_ij" line per stored entry
// (A.nonZerosInRow(i): five entries on an interior row, the diagonal alone on a Dirichlet row)
template <class SpMat>
void saveMatrixOnFile(SpMat A, const std::string &fileName)
{
    std::ofstream file(fileName, std::ofstream::trunc);
    file << A.rows() << " " << A.cols() << " " << A.nonZeros() << std::endl;
    for (size_t i = 0; i < A.rows(); i++) {
        const std::vector<size_t> row = A.nonZerosInRow(i);
        for (const auto &j : row) file << i << " " << j << " " << A.coeffRef(i, j) << std::endl;
    }
}

// first line = element count, then one value per line, default ostream precision
template <class Vector>
void saveVectorOnFile(const Vector &f, const std::string &fileName)
{
    std::ofstream file(fileName, std::ofstream::trunc);
    file << f.size() << std::endl;
    for (size_t i = 0; i < f.size(); i++) file << f[i] << std::endl;
}

}  // namespace Utils
#endif
