// fp8 (A format 0, B format 0; 0 = e4m3fn, 1 = e5m2) MFMA GEMM kernels, tile variant 0 (256, 256, 2, 4).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER(gnk_launch_f8_00_v0, 256, 256, 2, 4, 128, 128, 2, 2, OP_F8, 0, 0)
