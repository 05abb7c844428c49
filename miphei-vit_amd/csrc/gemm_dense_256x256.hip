#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_DENSE_UNIT(256, 256, 2, 4)
}
