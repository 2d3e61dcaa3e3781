// int8 MFMA GEMM kernels, tile variant 0 (256 x 256 four-stage ping-pong): 4 layouts x 3 epilogues.
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_PP2(gnk_launch_i8_v0, OP_I8, 0, 0)
