// gnk_gemm_w4.h -- 256 x 256 MFMA GEMM for 8-bit operands on gfx950 with ONE wave per SIMD.
//
// Round 3 design point.  The 8-wave ping-pong kernels (gnk_gemm_pp.h, gnk_gemm_pp2.h) put two waves on every
// SIMD and hand the matrix pipe from one to the other at a barrier 4-8 times per 128 k; their PMC shows the pipe busy 55-58 % of
// the dispatch.  Here a workgroup is 4 waves (2 x 2), each alone on its SIMD with a 128 x 128 wave tile:
//   * 256 accumulator registers per lane (the whole AGPR half of the 512-entry file), two complete fragment sets in VGPRs;
//   * LDS holds four 64-k stages of each operand (64 KiB + 64 KiB): an M/N-major operand as a ring of four [64 k][256 B] stages, a
//     K-major one as a ring of two 128-k super-stages [256 rows][128 B], so that every LDS-DMA piece fetches whole 128-byte lines
//     (with 64-byte row segments A@B.T ran 9 % behind A.T@B: twice the L2 requests); two to three stages of LDS-DMA are in flight;
//   * per stage a wave issues 64 x v_mfma_i32_16x16x64_i8 (or 16 x v_mfma_f32_32x32x64_f8f6f4) from fragment set `cur`
//     while the SAME wave, between those MFMAs, reads the next stage's fragments into set `nxt` (16 x ds_read_b128 per 64 MFMAs
//     instead of 12 per 32: a third less LDS traffic per MFMA than the 128 x 64 wave tile) and issues its 8 LDS-DMA pieces of
//     stage t + 4;
//   * ONE barrier per stage.  No MFMA ever waits for an LDS read issued in the same stage.
// Stage t, every wave:
//     vmcnt(n): my pieces of stage t+1 have landed (n from the issue order, tools/gen_w4_asm.py) | lgkmcnt(0): set `cur` (stage t) is in registers | s_barrier
//     MFMAs of stage t, and behind them: the reads of stage t+1 -> nxt (first half of the stage), the DMA pieces of stage t+4 -> buffer t & 3
// Buffer safety: buffer t & 3 (stage t) was read during stage t-1; every wave retired those reads (lgkmcnt(0)) before
// the barrier above, and the DMA that overwrites it is issued after that barrier.  Stage t+1 is read after the barrier that
// follows every wave's own vmcnt(n).  The loop is branch-free: past the end of K the base pointers stop advancing, so the surplus
// pieces re-fetch the last stage into a buffer nobody reads again (valid memory, L2 hits) and the vmcnt count never changes.
// The main loop is hand-scheduled inline asm (gnk_gemm_w4_asm.inc, written by tools/gen_w4_asm.py, which also states the fixed
// register plan): under hipcc the same program rotated its accumulators through ~100 v_accvgpr copies per four stages and ran
// 25 % SLOWER than the 8-phase kernel.  Images, staging maps and epilogues are those of gnk_gemm_pp2.h (two 128 x 64 halves per wave).
// Host-checked: K % 256 == 0 (four stages per loop iteration).
#pragma once
#include <atomic>

#include "gnk_gemm_pp2.h"
#include "gnk_gemm_w4_asm.inc"

#ifdef GNK_W4_STAMP
static __device__ unsigned long long gnk_w4_stamps[4096];   // one copy per instantiation unit; read back by gnk_debug_w4_stamps (gnk_gemm_i8_v5.hip)
#endif

namespace gnk_gemm {

// This wave's 1-KiB pieces of one operand, four waves per workgroup: four per 64-k stage of an M/N-major operand (256 B x 64 k), eight
// per 128-k super-stage of a K-major one (256 rows x 128 B: whole 128-byte lines per row, the image of the 8-phase kernel).
template <bool KMAJOR> struct StagerW4 {
    const char *base;
    unsigned off[8];
    int64_t step;
    __device__ __forceinline__ void init(const char *ptr, int64_t ld, int64_t r0, int64_t extent, int wave, int lane) {
        base = KMAJOR ? ptr + r0 * ld : ptr + r0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = (i * 4 + wave) * 64 + lane;
            if constexpr (KMAJOR) {
                int r, c;
                gnk_kmaj_stage_src(p, r, c);
                const int64_t rmax = extent - 1 - r0;
                const int64_t row = r < rmax ? r : rmax;
                off[i] = (unsigned)(row * ld + 16 * c);
            } else {
                int k, c;
                gnk_mn64_stage_src(p & 1023, k, c);
                const int64_t cmax = extent - 16 - r0;
                const int64_t col = 16 * c < cmax ? 16 * c : cmax;
                off[i] = (unsigned)((int64_t)k * ld + col);   // pieces 4-7 repeat 0-3 (unused)
            }
        }
        step = KMAJOR ? 128 : 64 * ld;
    }
};

// 16 accumulator registers a[16 N .. 16 N + 15] -> VGPRs (the asm loop leaves its results in the AGPR file)
template <int N> __device__ __forceinline__ void w4_read16(int (&r)[16]) {
#define GNK_W4_RDCASE(n)                                                                                                              \
    if constexpr (N == n)                                                                                                             \
        asm volatile(GNK_W4_RD_##n : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]),  \
                                     "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]));
    GNK_W4_RDCASE(0) GNK_W4_RDCASE(1) GNK_W4_RDCASE(2) GNK_W4_RDCASE(3) GNK_W4_RDCASE(4) GNK_W4_RDCASE(5) GNK_W4_RDCASE(6) GNK_W4_RDCASE(7)
    GNK_W4_RDCASE(8) GNK_W4_RDCASE(9) GNK_W4_RDCASE(10) GNK_W4_RDCASE(11) GNK_W4_RDCASE(12) GNK_W4_RDCASE(13) GNK_W4_RDCASE(14) GNK_W4_RDCASE(15)
#undef GNK_W4_RDCASE
}

// ---- int8: wave tile 128 x 128 = two halves h of 8 x 4 tiles of 16 x 16; tile (h, i, j) at a[4 (32 h + 4 i + j) ..]
template <class Cfg> struct W4I8 {
    typedef PP2I8<Cfg> Base;
    static __device__ __forceinline__ unsigned frag0(bool kmajor, int rb, int lane) {
        return kmajor ? (unsigned)gnk_kmaj_frag_offset(rb, 0, lane) : (unsigned)gnk_mn64_frag16_offset(rb, 0, lane);
    }
    template <int H, int I> static __device__ __forceinline__ void fetch_row(typename Base::acc_arr &acc) {
        int r[16];
        w4_read16<8 * H + I>(r);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[I][j] = (v4i){r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]};
    }
    template <int H> static __device__ __forceinline__ void finish_half(const GemmParams &p, char *smem, int wave, int lane, int64_t mwave, int64_t nwave,
                                                                        const float *lsc, int mlw, int nlw) {
        typename Base::acc_arr acc;
        fetch_row<H, 0>(acc); fetch_row<H, 1>(acc); fetch_row<H, 2>(acc); fetch_row<H, 3>(acc);
        fetch_row<H, 4>(acc); fetch_row<H, 5>(acc); fetch_row<H, 6>(acc); fetch_row<H, 7>(acc);
        Base::finish(p, acc, smem, wave * 2 + H, lane, mwave, nwave + 64 * H, lsc, mlw, nlw + 64 * H);
    }
};

// ---- fp8: two halves of 4 x 2 tiles of 32 x 32 (v_mfma_f32_32x32x64_f8f6f4); tile (h, i, j) at a[16 (8 h + 2 i + j) ..]
template <class Cfg> struct W4F8 {
    typedef PP2F8<Cfg> Base;
    static __device__ __forceinline__ unsigned frag0(bool kmajor, int rb, int lane) {
        return kmajor ? (unsigned)gnk_kmaj_offset(rb + (lane & 31), lane >> 5) : (unsigned)gnk_mn64_frag_offset(rb, 0, 0, lane);
    }
    template <int H, int T> static __device__ __forceinline__ void fetch_tile(typename Base::acc_arr &acc) {
        int r[16];
        w4_read16<8 * H + T>(r);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[T >> 1][T & 1][e] = __int_as_float(r[e]);
    }
    template <int H> static __device__ __forceinline__ void finish_half(const GemmParams &p, char *smem, int wave, int lane, int64_t mwave, int64_t nwave,
                                                                        const float *lsc, int mlw, int nlw) {
        typename Base::acc_arr acc;
        fetch_tile<H, 0>(acc); fetch_tile<H, 1>(acc); fetch_tile<H, 2>(acc); fetch_tile<H, 3>(acc);
        fetch_tile<H, 4>(acc); fetch_tile<H, 5>(acc); fetch_tile<H, 6>(acc); fetch_tile<H, 7>(acc);
        Base::finish(p, acc, smem, wave * 2 + H, lane, mwave, nwave + 64 * H, lsc, mlw, nlw + 64 * H);
    }
};

template <class Cfg, int OP> struct W4Select;
template <class Cfg> struct W4Select<Cfg, OP_I8> { typedef W4I8<Cfg> type; };
template <class Cfg> struct W4Select<Cfg, OP_F8> { typedef W4F8<Cfg> type; };

template <class Cfg>
__global__ void __launch_bounds__(256) gemm_w4_kernel(const GemmParams p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename W4Select<Cfg, Cfg::OP>::type K;

    // split-K (plain int32 output only), exactly as gemm_pp2_kernel
    GemmParams p = p_in;
    if constexpr (Cfg::EPI == EPI_PLAIN && Cfg::OP == OP_I8) {
        if (p.kchunk > 0) {
            const int64_t c0 = p.kchunk0 > 0 ? p.kchunk0 : p.kchunk;
            const int64_t k0 = blockIdx.y == 0 ? 0 : c0 + (int64_t)(blockIdx.y - 1) * p.kchunk;
            const int64_t len = blockIdx.y == 0 ? c0 : p.kchunk;
            const int64_t kn = p_in.K - k0 < len ? p_in.K - k0 : len;
            p.K = kn > 0 ? kn : 0;
            p.A += Cfg::A_KMAJOR ? k0 : k0 * p.lda;
            p.B += Cfg::B_KMAJOR ? k0 : k0 * p.ldb;
            p.C = (char *)p.C + (int64_t)blockIdx.y * p.M * p.N * 4;
        }
    }

#ifdef GNK_W4_STAMP
    const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tm, tn;
    tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

    // row / column scales of the block tile: two per thread, parked in LDS after the K loop (see ScalePrefetch)
    unsigned raw_r = 0, raw_c = 0;
    if constexpr (Cfg::EPI == EPI_ROWCOL) {
        const int t = threadIdx.x;
        const int64_t mi = m0 + t < p.M - 1 ? m0 + t : p.M - 1;
        const int64_t ni = p.col_scalar ? 0 : (n0 + t < p.N - 1 ? n0 + t : p.N - 1);
        raw_r = gnk_load_raw_rt(p.s0, p.scale_dtype, mi);
        raw_c = gnk_load_raw_rt(p.s1, p.scale_dtype, ni);
    }

    StagerW4<Cfg::A_KMAJOR> sa;
    StagerW4<Cfg::B_KMAJOR> sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);

    const int ns = (int)(p.K / Cfg::SK);   // a multiple of 4, >= 4 (host-checked)
    const unsigned lds0 = (unsigned)(uintptr_t)(GNK_LDS char *)smem;
    const unsigned va = lds0 + K::frag0(Cfg::A_KMAJOR, wm * 128, lane);
    const unsigned vb = lds0 + 4 * Cfg::OPER_BYTES + K::frag0(Cfg::B_KMAJOR, wn * 128, lane);
    const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sa.base), ahi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)sa.base >> 32));
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sb.base), bhi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)sb.base >> 32));
    const unsigned stepa = __builtin_amdgcn_readfirstlane((unsigned)sa.step), stepb = __builtin_amdgcn_readfirstlane((unsigned)sb.step);
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    const int nit = __builtin_amdgcn_readfirstlane(ns >> 2), rem = __builtin_amdgcn_readfirstlane(ns - 1);

#define GNK_W4_RUN(TEXT)                                                                                                              \
    asm volatile(TEXT : : [alo] "s"(alo), [ahi] "s"(ahi), [blo] "s"(blo), [bhi] "s"(bhi), [stepa] "s"(stepa), [stepb] "s"(stepb),      \
                 [lds] "s"(ldsw), [nit] "s"(nit), [rem] "s"(rem), [va] "v"(va), [vb] "v"(vb),                                         \
                 [oa0] "v"(sa.off[0]), [oa1] "v"(sa.off[1]), [oa2] "v"(sa.off[2]), [oa3] "v"(sa.off[3]),                              \
                 [oa4] "v"(sa.off[4]), [oa5] "v"(sa.off[5]), [oa6] "v"(sa.off[6]), [oa7] "v"(sa.off[7]),                              \
                 [ob0] "v"(sb.off[0]), [ob1] "v"(sb.off[1]), [ob2] "v"(sb.off[2]), [ob3] "v"(sb.off[3]),                              \
                 [ob4] "v"(sb.off[4]), [ob5] "v"(sb.off[5]), [ob6] "v"(sb.off[6]), [ob7] "v"(sb.off[7]),                              \
                 [fa] "i"(Cfg::FA), [fb] "i"(Cfg::FB) : GNK_W4_CLOBBERS)
    if constexpr (Cfg::OP == OP_I8) {
        if constexpr (Cfg::A_KMAJOR && Cfg::B_KMAJOR) GNK_W4_RUN(GNK_W4_ASM_I8_KK);
        else if constexpr (Cfg::A_KMAJOR) GNK_W4_RUN(GNK_W4_ASM_I8_KM);
        else if constexpr (Cfg::B_KMAJOR) GNK_W4_RUN(GNK_W4_ASM_I8_MK);
        else GNK_W4_RUN(GNK_W4_ASM_I8_MM);
    } else {
        if constexpr (Cfg::A_KMAJOR && Cfg::B_KMAJOR) GNK_W4_RUN(GNK_W4_ASM_F8_KK);
        else if constexpr (Cfg::A_KMAJOR) GNK_W4_RUN(GNK_W4_ASM_F8_KM);
        else if constexpr (Cfg::B_KMAJOR) GNK_W4_RUN(GNK_W4_ASM_F8_MK);
        else GNK_W4_RUN(GNK_W4_ASM_F8_MM);
    }
#undef GNK_W4_RUN
    // the asm ends behind vmcnt(0) + a barrier: every DMA has landed and every wave is done reading, LDS is free for the epilogue
#ifdef GNK_W4_STAMP   // diagnostic build (tools/w4_ablate.sh 16): s_memtime / s_memrealtime around the loop, one record per workgroup
    unsigned stamp_t[8];
    asm volatile("s_mov_b32 %0, s82\n\ts_mov_b32 %1, s83\n\ts_mov_b32 %2, s84\n\ts_mov_b32 %3, s85\n\ts_mov_b32 %4, s86\n\ts_mov_b32 %5, s87\n\t"
                 "s_mov_b32 %6, s88\n\ts_mov_b32 %7, s89" : "=s"(stamp_t[0]), "=s"(stamp_t[1]), "=s"(stamp_t[2]), "=s"(stamp_t[3]), "=s"(stamp_t[4]),
                 "=s"(stamp_t[5]), "=s"(stamp_t[6]), "=s"(stamp_t[7]));
#endif

    float *const lsc = (float *)(smem + Cfg::LDS_BYTES);
    if constexpr (Cfg::EPI == EPI_ROWCOL) {
        *(GNK_LDS float *)(GNK_LDS char *)((char *)lsc + 4 * threadIdx.x) = gnk_raw_to_f32_rt(raw_r, p.scale_dtype);
        *(GNK_LDS float *)(GNK_LDS char *)((char *)lsc + 4 * (256 + threadIdx.x)) = gnk_raw_to_f32_rt(raw_c, p.scale_dtype);
    }
    __syncthreads();
    const float *const sc = Cfg::EPI == EPI_ROWCOL ? (const float *)lsc : nullptr;
    K::template finish_half<0>(p, smem, wave, lane, m0 + wm * 128, n0 + wn * 128, sc, wm * 128, wn * 128);
#ifdef GNK_W4_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long stamp_half = __builtin_amdgcn_s_memrealtime();
#endif
    K::template finish_half<1>(p, smem, wave, lane, m0 + wm * 128, n0 + wn * 128, sc, wm * 128, wn * 128);
#ifdef GNK_W4_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long stamp_end = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 512) {   // record: cycles before / after the loop, then 100 MHz ticks: entry, loop start, loop end, half 0 stored, all stored
        unsigned long long *o = gnk_w4_stamps + blockIdx.x * 8;
        o[0] = ((unsigned long long)stamp_t[1] << 32) | stamp_t[0];
        o[1] = ((unsigned long long)stamp_t[5] << 32) | stamp_t[4];
        o[2] = stamp_entry;
        o[3] = ((unsigned long long)stamp_t[3] << 32) | stamp_t[2];
        o[4] = ((unsigned long long)stamp_t[7] << 32) | stamp_t[6];
        o[5] = stamp_half;
        o[6] = stamp_end;
    }
#endif
}

template <class Cfg> int launch_w4(const GemmParams &p0, hipStream_t st) {
    constexpr int LDS_TOTAL = Cfg::LDS_BYTES + (Cfg::EPI == EPI_ROWCOL ? 2048 : 0);
    GemmParams p = p0;
    p.grid_m = (int)((p.M + 255) / 256);
    p.grid_n = (int)((p.N + 255) / 256);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "gemm: no current HIP device");
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void *)gemm_w4_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return gnk_fail(GNK_ERR_LAUNCH, "gemm: cannot reserve %d bytes of LDS", LDS_TOTAL);
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int64_t c0 = p.kchunk0 > 0 ? p.kchunk0 : p.kchunk;
    const unsigned splits = p.kchunk > 0 ? 1u + (unsigned)((p.K - c0 + p.kchunk - 1) / p.kchunk) : 1u;
    hipLaunchKernelGGL((gemm_w4_kernel<Cfg>), dim3((unsigned)(p.grid_m * p.grid_n), splits), dim3(256), LDS_TOTAL, st, p);
    return gnk_check_launch("gemm (one wave per SIMD)");
}

}  // namespace gnk_gemm
