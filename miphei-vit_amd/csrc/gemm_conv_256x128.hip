#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_CONV_UNIT(256, 128, 4, 2)
}
