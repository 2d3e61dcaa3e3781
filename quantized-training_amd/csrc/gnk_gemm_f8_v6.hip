// fp8 MFMA GEMM kernels, tile variant 6 (64 x 128 tile, see gnk_gemm_i8_v6.hip): 4 format pairs x 4 layouts x 2 epilogues.
#include "gnk_gemm_dispatch.h"

template <int FA, int FB, bool AK, bool BK> static int launch_64x128(int epi, const GemmParams &p, hipStream_t st) {
    switch (epi) {
    case EPI_PLAIN: return gnk_gemm::launch<GemmCfg<64, 128, 2, 2, AK, BK, OP_F8, FA, FB, EPI_PLAIN>>(p, st);
    case EPI_ROWCOL: return gnk_gemm::launch<GemmCfg<64, 128, 2, 2, AK, BK, OP_F8, FA, FB, EPI_ROWCOL>>(p, st);
    default: return gnk_fail(GNK_ERR_UNSUPPORTED, "tile-scaled products run on the 128 x 128 kernels (variant 1)");
    }
}

template <int FA, int FB> static int by_layout(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {
    if (a_kmajor) return b_kmajor ? launch_64x128<FA, FB, true, true>(epi, p, st) : launch_64x128<FA, FB, true, false>(epi, p, st);
    return b_kmajor ? launch_64x128<FA, FB, false, true>(epi, p, st) : launch_64x128<FA, FB, false, false>(epi, p, st);
}

int gnk_launch_f8_v6(int fa, int fb, bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {
    if (fa == F8_E4M3) return fb == F8_E4M3 ? by_layout<0, 0>(a_kmajor, b_kmajor, epi, p, st) : by_layout<0, 1>(a_kmajor, b_kmajor, epi, p, st);
    return fb == F8_E4M3 ? by_layout<1, 0>(a_kmajor, b_kmajor, epi, p, st) : by_layout<1, 1>(a_kmajor, b_kmajor, epi, p, st);
}
