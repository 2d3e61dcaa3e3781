// fp8 (A format 0, B format 0; 0 = e4m3fn, 1 = e5m2) MFMA GEMM kernels, tile variant 5 (one wave per SIMD, gnk_gemm_w4.h).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_W4(gnk_launch_f8_00_v5, OP_F8, 0, 0)
