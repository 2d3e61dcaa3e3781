// fp8 (e4m3fn x e4m3fn) MFMA GEMM kernels, tile variant 4: 256 x 256 8-phase ping-pong over two 128-k buffers, one
// v_mfma_scale_f32_16x16x128_f8f6f4 per tile and K-step (the fp8 shape that sustains the highest rate on this chip).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_PP8(gnk_launch_f8_00_v4, OP_F8, 0, 0)
