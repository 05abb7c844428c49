#include "gemm_kernel.hpp"
namespace mvit_gemm {
MVIT_GEMM_CONV_UNIT(128, 128, 2, 2)
}
