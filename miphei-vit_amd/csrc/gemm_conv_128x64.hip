#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_CONV_UNIT(128, 64, 2, 2)
}
