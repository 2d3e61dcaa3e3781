// int8 MFMA GEMM kernels, tile variant 4: 256 x 256 8-phase ping-pong over two 128-k buffers (kept for A/B timing).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_PP8(gnk_launch_i8_v4, OP_I8, 0, 0)
