// fp8 (A format 0, B format 1; 0 = e4m3fn, 1 = e5m2) MFMA GEMM kernels, tile variant 1 (128, 128, 2, 2).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER(gnk_launch_f8_01_v1, 128, 128, 2, 2, 128, 128, 2, 2, OP_F8, 0, 1)
