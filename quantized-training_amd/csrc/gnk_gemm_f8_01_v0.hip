// fp8 (A format 0, B format 1; 0 = e4m3fn, 1 = e5m2) MFMA GEMM kernels, tile variant 0 (256 x 256 four-stage ping-pong).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_PP2(gnk_launch_f8_01_v0, OP_F8, 0, 1)
