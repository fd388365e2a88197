// utilities.cpp -- see utilities.hpp.  Behavioural contract: flags
// -n -a -w -ml -test -smt --help, defaults, echo lines and `Error: …` messages on stdout
// followed by exit(1), as in /root/reference/GeometricMultigrid/src/utilities.cpp:3-159.
#include "utilities.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>

namespace {

[[noreturn]] void fail(const char *msg)
{
    std::cout << "Error: " << msg << std::endl;
    std::exit(1);
}

bool parse_int(const char *s, long &out)
{
    char *end = nullptr;
    out = std::strtol(s, &end, 10);
    return end != s;
}

void usage()
{
    std::cout << "Usage: ./Multigrid [OPTIONS]\n" << std::endl
              << "Options:" << std::endl
              << "  -n, insert number of spaces" << std::endl
              << "  -a, specifies differential constant" << std::endl
              << "  -w, insert the Width of the rectangle domain" << std::endl
              << "  -ml, insert multigrid level" << std::endl
              << "  -test, insert type of function in input to test it" << std::endl
              << "  -smt, you can choose your favourite smoother" << std::endl
              << "  --help, Display this help message" << std::endl
              << "MI355X extensions:" << std::endl
              << "  -dim 2|3, -cycle saw|v, -omega W, -nu1 K, -nu2 K, -rbgs, -zebra, -zebrax, -anisox A, -anisoy A, -fw, -coarse_fixed K, -fp32, -maxit K, -cold, -eps E, -semi K" << std::endl;
}

}  // namespace

void Utils::parse_command_line(int argc, char **argv, Options &o)
{
    if (argc < 2) {
        std::cout << "Inserted by default N = " << DEFAULT_N << std::endl;
        std::cout << "Inserted by default alpha = " << DEFAULT_ALPHA << std::endl;
        std::cout << "Inserted by default width = " << DEFAULT_WIDTH << std::endl;
        std::cout << "Inserted by default multigrid level = " << DEFAULT_LEVEL << std::endl;
        std::cout << "Inserted by default test number " << DEFAULT_TEST << std::endl;
        std::cout << "Inserted by default Smooter number " << DEFAULT_METHOD << std::endl;
        return;
    }
    for (int i = 0; i < argc; i++) {
        const std::string a = argv[i];
        const bool has_value = i + 1 < argc;
        long v = 0;
        if (a == "--help") {
            usage();
            std::exit(1);
        } else if (a == "-n") {
            if (!has_value) fail("Please, insert something");
            if (!parse_int(argv[i + 1], v)) fail("Please, insert a number after -n");
            o.N = static_cast<size_t>(v);
            std::cout << "Inserted N = " << o.N << std::endl;
            if (v <= 0) fail("Please, insert a valid N value");
        } else if (a == "-a" && has_value) {
            o.alpha = std::atof(argv[i + 1]);
            std::cout << "Inserted alpha = " << o.alpha << std::endl;
        } else if (a == "-ml" && has_value) {
            if (!parse_int(argv[i + 1], v)) fail("Please, insert a number after -ml");
            o.level = static_cast<int>(v);
            std::cout << "Inserted level = " << o.level << std::endl;
            if (v <= 0) fail("Please, insert a valid level");
        } else if (a == "-smt" && has_value) {
            if (!parse_int(argv[i + 1], v)) fail("Please, insert a number after -smt");
            // (the reference casts first and range-checks the enum afterwards -- undefined behaviour for a value
            // outside the enumeration, flagged by UBSan; same echo line and same fallback here without it)
            std::cout << "Inserted Smoother number = " << v << std::endl;
            o.smoother = (v >= 0 && v < SMOOTHERS_END) ? static_cast<SMOOTHERS>(v) : DEFAULT_METHOD;
        } else if (a == "-test" && has_value) {
            if (!parse_int(argv[i + 1], v)) fail("Please, insert a double after -test");
            o.test = static_cast<int>(v);
            std::cout << "Inserted test number = " << o.test << std::endl;
            if (v < 0) fail("Please, insert a valid test number");
        } else if (a == "-w" && has_value) {
            o.width = std::atof(argv[i + 1]);
            std::cout << "Inserted width = " << o.width << std::endl;
            if (o.width <= 0) fail("Please, insert a valid width");
        }
        // ---- extensions ----
        else if (a == "-dim" && has_value) { o.dim = std::atoi(argv[i + 1]); if (o.dim != 2 && o.dim != 3) fail("-dim must be 2 or 3"); }
        else if (a == "-cycle" && has_value) { o.vcycle = std::string(argv[i + 1]) == "v"; }
        else if (a == "-omega" && has_value) { o.omega = std::atof(argv[i + 1]); }
        else if (a == "-nu1" && has_value) { o.nu1 = std::atoi(argv[i + 1]); }
        else if (a == "-nu2" && has_value) { o.nu2 = std::atoi(argv[i + 1]); }
        else if (a == "-coarse_fixed" && has_value) { o.coarse_fixed = std::atoi(argv[i + 1]); }
        else if (a == "-maxit" && has_value) { o.maxit = std::atoi(argv[i + 1]); if (o.maxit < 0) fail("Please, insert a valid -maxit value"); }
        else if (a == "-eps" && has_value) { o.eps_z = std::atof(argv[i + 1]); }
        else if (a == "-semi" && has_value) { o.semi = std::atoi(argv[i + 1]); }
        else if (a == "-rbgs") { o.rbgs = true; }
        else if (a == "-zebra") { o.zebra = true; }  // zebra line Gauss-Seidel along y
        else if (a == "-zebrax") { o.zebrax = true; }  // ... along x
        else if (a == "-anisox" && has_value) { o.aniso_x = std::atof(argv[i + 1]); }
        else if (a == "-anisoy" && has_value) { o.aniso_y = std::atof(argv[i + 1]); }
        else if (a == "-fw") { o.full_weighting = true; }
        else if (a == "-fp32") { o.fp32 = true; }
        else if (a == "-cold") { o.cold = true; }
    }
}

void Utils::Initialization_for_N(int argc, char **argv, size_t &N, double &alpha, double &width, int &level,
                                 int &functions_to_test, SMOOTHERS &sm)
{
    Options o;
    parse_command_line(argc, argv, o);
    N = o.N; alpha = o.alpha; width = o.width; level = o.level; functions_to_test = o.test; sm = o.smoother;
}

void Utils::init_test_functions(std::function<double(const double, const double)> &f,
                                std::function<double(const double, const double)> &g, int i)
{
    using fn = std::function<double(const double, const double)>;
    // even entries: forcing f, odd entries: Dirichlet data g.  The marker comments keep the
    // reference web form's scraper (WebInterface/FuncHandle.php:19-26) working on this file.
    const fn table[6] = {
        // FFF
        [](const double, const double) { return 1.; },
        [](const double, const double) { return 0.; },
        [](const double x, const double y) { return -5.0 * exp(x) * exp(-2.0 * y); },
        [](const double x, const double y) { return exp(x) * exp(-2.0 * y); },
        [](const double x, const double y) { double r = std::sqrt(x * x + y * y); return r != 0.0 ? -30. * (std::cos(30. * r) / r - 30. * std::sin(30. * r)) : 0.0; },
        [](const double x, const double y) { return std::sin(30. * std::sqrt(x * x + y * y)); }
        // END
    };
    if (i < 0 || i > 2) {
        i = 0;
        std::cout << "Warning: Invalid test case index. Default test case selected.\n";
    }
    f = table[2 * i];
    g = table[2 * i + 1];
}
