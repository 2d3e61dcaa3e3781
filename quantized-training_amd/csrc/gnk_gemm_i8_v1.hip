// int8 MFMA GEMM kernels, tile variant 1 (128, 128, 2, 2): 4 layouts x 3 epilogues.
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER(gnk_launch_i8_v1, 128, 128, 2, 2, 128, 128, 2, 2, OP_I8, 0, 0)
