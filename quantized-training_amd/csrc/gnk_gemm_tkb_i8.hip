// Tile-scaled MFMA GEMM with 64- / 32-k scale blocks, int8: 4 layouts x 2 block sizes.
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_TKB(gnk_launch_tkb_i8, OP_I8, 0, 0)
