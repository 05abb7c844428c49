#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_DENSE_UNIT(128, 32, 4, 1)
}
