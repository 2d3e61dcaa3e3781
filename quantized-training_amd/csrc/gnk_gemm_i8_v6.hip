// int8 MFMA GEMM kernels, tile variant 6: 64 x 128 tile over 4 waves (32 x 64 wave tiles), two-buffer loop: 4 layouts x 2 epilogues.
#include "gnk_gemm_dispatch.h"

template <bool AK, bool BK> static int launch_64x128(int epi, const GemmParams &p, hipStream_t st) {
    switch (epi) {
    case EPI_PLAIN: return gnk_gemm::launch<GemmCfg<64, 128, 2, 2, AK, BK, OP_I8, 0, 0, EPI_PLAIN>>(p, st);
    case EPI_ROWCOL: return gnk_gemm::launch<GemmCfg<64, 128, 2, 2, AK, BK, OP_I8, 0, 0, EPI_ROWCOL>>(p, st);
    default: return gnk_fail(GNK_ERR_UNSUPPORTED, "tile-scaled products run on the 128 x 128 kernels (variant 1)");
    }
}

int gnk_launch_i8_v6(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {
    if (a_kmajor) return b_kmajor ? launch_64x128<true, true>(epi, p, st) : launch_64x128<true, false>(epi, p, st);
    return b_kmajor ? launch_64x128<false, true>(epi, p, st) : launch_64x128<false, false>(epi, p, st);
}
