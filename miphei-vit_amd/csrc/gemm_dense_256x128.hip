#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_DENSE_UNIT(256, 128, 4, 2)
}
