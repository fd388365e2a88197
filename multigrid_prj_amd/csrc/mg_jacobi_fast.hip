// mg_jacobi_fast.hip -- finest-grid fast paths (filled in after microbenchmarks).
#include "mg_kernels.h"
namespace mg {
}
