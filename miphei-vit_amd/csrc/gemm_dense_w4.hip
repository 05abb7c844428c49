#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_DENSE_UNIT(256, 256, 2, 2)
MVIT_GEMM_DENSE_UNIT(256, 128, 2, 2)
}
