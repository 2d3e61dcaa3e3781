// fp8 (A format 0, B format 1; 0 = e4m3fn, 1 = e5m2) MFMA GEMM kernels, tile variant 4 (256 x 256 8-phase ping-pong, 16x16x128 MFMA).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_PP8(gnk_launch_f8_01_v4, OP_F8, 0, 1)
