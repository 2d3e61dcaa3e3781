// Tile-scaled MFMA GEMM with 64- / 32-k scale blocks, fp8 e5m2 x e4m3: 4 layouts x 2 block sizes.
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_TKB(gnk_launch_tkb_f8_10, OP_F8, 1, 0)
