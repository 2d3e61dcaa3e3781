// int8 MFMA GEMM kernels, tile variant 3: 256 x 256 two-buffer loop (kept for A/B timing against the ping-pong kernel).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER(gnk_launch_i8_v3, 256, 256, 2, 4, 128, 128, 2, 2, OP_I8, 0, 0)
