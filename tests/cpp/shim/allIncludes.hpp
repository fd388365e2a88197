// Test shim: lets a translation unit written against the reference's umbrella header
// (`#include "allIncludes.hpp"`) compile against the MI355X mirror instead.
#ifndef ALL_H
#define ALL_H
#include <chrono>
#include <cmath>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <numeric>
#include <random>
#include <tuple>
#include <vector>

#include "utilities.hpp"
#include "multigrid_hip.hpp"
#endif
