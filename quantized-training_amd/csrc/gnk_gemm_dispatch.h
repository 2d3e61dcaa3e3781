// gnk_gemm_dispatch.h -- launcher table shared by the instantiation units and gnk_api.hip.
#pragma once
#include "gnk_gemm_kernel.h"
#include "gnk_gemm_pp.h"
#include "gnk_gemm_pp2.h"
#include "gnk_gemm_w4.h"

// GEMM tile variants (gnk_set_gemm_variant):
//   0: 256 x 256 tile, 8 waves, four 64-k stages, 2-phase ping-pong (gnk_gemm_pp2.h), 1 workgroup / CU -- large problems
//   1: 128 x 128 tile, 4 waves (2 x 2), two-buffer loop, 64 KiB LDS, 2 workgroups / CU    -- mid-size problems
//   2: generic 64 x 64 kernel (any strides / sizes; the fallback)
//   3: 256 x 256 tile, 8 waves, two-buffer loop (the first kernel; int8 only, kept for A/B timing)
//   4: 256 x 256 tile, 8 waves, two 128-k buffers, 8-phase ping-pong (gnk_gemm_pp.h): int8, and fp8 on the 16x16x128 MFMA;
//      the heuristic's choice whenever A is K-major (A@B.T, A@B)
//   5: 256 x 256 tile, 4 waves (one per SIMD, 128 x 128 wave tile), four 64-k stages, one barrier per stage (gnk_gemm_w4.h)
//   6: 64 x 128 tile, 4 waves (2 x 2, 32 x 64 wave tiles), two-buffer loop, 48 KiB LDS: int8 / fp8 plain / row-col products whose 128 x 128 tiling
//      would put at most one workgroup on a CU (mid sizes, grad-weights of narrow Linears) -- twice the workgroups, two instruction streams per SIMD
enum { GNK_VARIANT_256 = 0, GNK_VARIANT_128 = 1, GNK_VARIANT_GENERIC = 2, GNK_VARIANT_256_SIMPLE = 3, GNK_VARIANT_256_PP8 = 4, GNK_VARIANT_256_W4 = 5,
       GNK_VARIANT_64X128 = 6, GNK_VARIANT_LAST = 6,
       GNK_VARIANT_AUTO_NO_W4 = -2 /* the heuristic without variant 5 (A/B timing) */, GNK_VARIANT_AUTO_NO_64 = -3 /* ... without variant 6 */ };

// fp8 format codes as the MFMA's cbsz/blgp fields want them
enum { F8_E4M3 = 0, F8_E5M2 = 1 };

// One launcher per (operand class, fp8 formats, tile variant): picks layout and epilogue at run time.
typedef int (*gnk_gemm_launcher)(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st);

#define GNK_DECLARE_LAUNCHER(NAME) int NAME(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st)

GNK_DECLARE_LAUNCHER(gnk_launch_i8_v0);
GNK_DECLARE_LAUNCHER(gnk_launch_i8_v1);
GNK_DECLARE_LAUNCHER(gnk_launch_i8_v3);
GNK_DECLARE_LAUNCHER(gnk_launch_i8_v4);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_00_v4);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_01_v4);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_10_v4);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_11_v4);
GNK_DECLARE_LAUNCHER(gnk_launch_i8_v5);
GNK_DECLARE_LAUNCHER(gnk_launch_i8_v6);
int gnk_launch_f8_v6(int fa, int fb, bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_00_v5);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_01_v5);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_10_v5);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_11_v5);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_00_v0);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_01_v0);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_10_v0);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_11_v0);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_00_v1);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_01_v1);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_10_v1);
GNK_DECLARE_LAUNCHER(gnk_launch_f8_11_v1);

// Defines NAME for the given tile shape / operand class: 4 layouts x 3 epilogues = 12 kernels.
// The tile-scaled epilogue keeps a second (fp32) accumulator set, so it gets its own, smaller
// wave tile (TBM x TBN over TWM x TWN waves).
#define GNK_DEFINE_LAUNCHER(NAME, BM, BN, WM, WN, TBM, TBN, TWM, TWN, OP, FA, FB)                                               \
    template <bool AK, bool BK> static int NAME##_l(int epi, const GemmParams &p, hipStream_t st) {         \
        switch (epi) {                                                                                      \
        case EPI_PLAIN: return gnk_gemm::launch<GemmCfg<BM, BN, WM, WN, AK, BK, OP, FA, FB, EPI_PLAIN>>(p, st);   \
        case EPI_ROWCOL: return gnk_gemm::launch<GemmCfg<BM, BN, WM, WN, AK, BK, OP, FA, FB, EPI_ROWCOL>>(p, st); \
        default:                                                                                            \
            if (p.QN % 4 != 0)   /* per-column B scales (GemmCfg::PERCOL) */                                    \
                return p.fused_rescale ? gnk_gemm::launch<GemmCfg<TBM, TBN, TWM, TWN, AK, BK, OP, FA, FB, EPI_TILE, GNK_BK, true, true>>(p, st)   \
                                       : gnk_gemm::launch<GemmCfg<TBM, TBN, TWM, TWN, AK, BK, OP, FA, FB, EPI_TILE, GNK_BK, false, true>>(p, st); \
            return p.fused_rescale ? gnk_gemm::launch<GemmCfg<TBM, TBN, TWM, TWN, AK, BK, OP, FA, FB, EPI_TILE, GNK_BK, true>>(p, st)   \
                                   : gnk_gemm::launch<GemmCfg<TBM, TBN, TWM, TWN, AK, BK, OP, FA, FB, EPI_TILE>>(p, st);            \
        }                                                                                                   \
    }                                                                                                       \
    int NAME(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {                  \
        if (a_kmajor) return b_kmajor ? NAME##_l<true, true>(epi, p, st) : NAME##_l<true, false>(epi, p, st); \
        return b_kmajor ? NAME##_l<false, true>(epi, p, st) : NAME##_l<false, false>(epi, p, st);           \
    }

// Variant 0: 4-stage ping-pong 256 x 256 kernel for the plain / row-col epilogues, 128 x 128 two-buffer
// kernel for the tile-scaled one.
#define GNK_DEFINE_LAUNCHER_PP2(NAME, OP, FA, FB)                                                           \
    template <bool AK, bool BK> static int NAME##_l(int epi, const GemmParams &p, hipStream_t st) {         \
        switch (epi) {                                                                                      \
        case EPI_PLAIN: return gnk_gemm::launch_pp2<gnk_gemm::PP2Cfg<AK, BK, OP, FA, FB, EPI_PLAIN>>(p, st);   \
        case EPI_ROWCOL: return gnk_gemm::launch_pp2<gnk_gemm::PP2Cfg<AK, BK, OP, FA, FB, EPI_ROWCOL>>(p, st); \
        default: return gnk_fail(GNK_ERR_UNSUPPORTED, "tile-scaled products run on the 128 x 128 kernels (variant 1)");   \
        }                                                                                                   \
    }                                                                                                       \
    int NAME(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {                  \
        if (a_kmajor) return b_kmajor ? NAME##_l<true, true>(epi, p, st) : NAME##_l<true, false>(epi, p, st); \
        return b_kmajor ? NAME##_l<false, true>(epi, p, st) : NAME##_l<false, false>(epi, p, st);           \
    }

// Variant 4: the 8-phase ping-pong kernel (two 128-k buffers).
#define GNK_DEFINE_LAUNCHER_PP8(NAME, OP, FA, FB)                                                           \
    template <bool AK, bool BK> static int NAME##_l(int epi, const GemmParams &p, hipStream_t st) {         \
        switch (epi) {                                                                                      \
        case EPI_PLAIN: return gnk_gemm::launch_pp<GemmCfg<256, 256, 2, 4, AK, BK, OP, FA, FB, EPI_PLAIN>>(p, st);   \
        case EPI_ROWCOL: return gnk_gemm::launch_pp<GemmCfg<256, 256, 2, 4, AK, BK, OP, FA, FB, EPI_ROWCOL>>(p, st); \
        default: return gnk_fail(GNK_ERR_UNSUPPORTED, "tile-scaled products run on the 128 x 128 kernels (variant 1)");   \
        }                                                                                                   \
    }                                                                                                       \
    int NAME(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {                  \
        if (a_kmajor) return b_kmajor ? NAME##_l<true, true>(epi, p, st) : NAME##_l<true, false>(epi, p, st); \
        return b_kmajor ? NAME##_l<false, true>(epi, p, st) : NAME##_l<false, false>(epi, p, st);           \
    }

// Variant 5: one wave per SIMD (gnk_gemm_w4.h).
#define GNK_DEFINE_LAUNCHER_W4(NAME, OP, FA, FB)                                                            \
    template <bool AK, bool BK> static int NAME##_l(int epi, const GemmParams &p, hipStream_t st) {         \
        switch (epi) {                                                                                      \
        case EPI_PLAIN: return gnk_gemm::launch_w4<gnk_gemm::PP2Cfg<AK, BK, OP, FA, FB, EPI_PLAIN>>(p, st);   \
        case EPI_ROWCOL: return gnk_gemm::launch_w4<gnk_gemm::PP2Cfg<AK, BK, OP, FA, FB, EPI_ROWCOL>>(p, st); \
        default: return gnk_fail(GNK_ERR_UNSUPPORTED, "tile-scaled products run on the 128 x 128 kernels (variant 1)");   \
        }                                                                                                   \
    }                                                                                                       \
    int NAME(bool a_kmajor, bool b_kmajor, int epi, const GemmParams &p, hipStream_t st) {                  \
        if (a_kmajor) return b_kmajor ? NAME##_l<true, true>(epi, p, st) : NAME##_l<true, false>(epi, p, st); \
        return b_kmajor ? NAME##_l<false, true>(epi, p, st) : NAME##_l<false, false>(epi, p, st);           \
    }

// Tile-scaled GEMM with 64- or 32-k scale blocks (Jetfire-style: 32-k MFMAs) or blocks of a multiple of 128 k above 128 (QK == 128
// stays with the launchers above): instances of the 128 x 128 two-buffer kernel, one instantiation unit per operand class
// (gnk_gemm_tkb_*.hip).
#define GNK_DECLARE_TKB(NAME) int NAME(int kb, bool a_kmajor, bool b_kmajor, const GemmParams &p, hipStream_t st)
GNK_DECLARE_TKB(gnk_launch_tkb_i8);
GNK_DECLARE_TKB(gnk_launch_tkb_f8_00);
GNK_DECLARE_TKB(gnk_launch_tkb_f8_01);
GNK_DECLARE_TKB(gnk_launch_tkb_f8_10);
GNK_DECLARE_TKB(gnk_launch_tkb_f8_11);

#define GNK_DEFINE_TKB(NAME, OP, FA, FB)                                                                                  \
    template <int KB, bool FU> static int NAME##_k(bool ak, bool bk, const GemmParams &p, hipStream_t st) {               \
        if (ak) return bk ? gnk_gemm::launch<GemmCfg<128, 128, 2, 2, true, true, OP, FA, FB, EPI_TILE, KB, FU>>(p, st)    \
                          : gnk_gemm::launch<GemmCfg<128, 128, 2, 2, true, false, OP, FA, FB, EPI_TILE, KB, FU>>(p, st);  \
        return bk ? gnk_gemm::launch<GemmCfg<128, 128, 2, 2, false, true, OP, FA, FB, EPI_TILE, KB, FU>>(p, st)           \
                  : gnk_gemm::launch<GemmCfg<128, 128, 2, 2, false, false, OP, FA, FB, EPI_TILE, KB, FU>>(p, st);         \
    }                                                                                                                     \
    template <bool FU> static int NAME##_f(int kb, bool a_kmajor, bool b_kmajor, const GemmParams &p, hipStream_t st) {   \
        if (kb > 128) return NAME##_k<256, FU>(a_kmajor, b_kmajor, p, st);   /* any multiple of 128: p.kgroup steps per block */  \
        return kb == 64 ? NAME##_k<64, FU>(a_kmajor, b_kmajor, p, st) : NAME##_k<32, FU>(a_kmajor, b_kmajor, p, st);      \
    }                                                                                                                     \
    int NAME(int kb, bool a_kmajor, bool b_kmajor, const GemmParams &p, hipStream_t st) {                                 \
        return p.fused_rescale ? NAME##_f<true>(kb, a_kmajor, b_kmajor, p, st) : NAME##_f<false>(kb, a_kmajor, b_kmajor, p, st);  \
    }

// ---------------------------------------------------------------- generic fallback (gnk_gemm_generic.hip)
struct GenericParams {
    const char *A;
    const char *B;
    void *C;
    const void *s0;
    const void *s1;
    int64_t M, N, K;
    int64_t sam, sak, sbk, sbn;
    int64_t ssam, ssak, ssbk, ssbn;
    int64_t QM, QN, QK;
    int epi, out_dtype, scale_dtype, col_scalar, order;
};
int gnk_launch_generic(int op, int fa, int fb, const GenericParams &p, hipStream_t st);
