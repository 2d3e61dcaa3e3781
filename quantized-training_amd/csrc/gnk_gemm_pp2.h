// gnk_gemm_pp2.h -- 256 x 256 four-stage "ping-pong" MFMA GEMM for 8-bit operands on gfx950.
//
// Measured on MI355X (tools/ablate*.py) the 8-phase kernel of gnk_gemm_pp.h spends, per 128 k:
// 0.87-0.99 us in MFMAs, +0.25 us in its 8 barriers, +0.3 us waiting for LDS-DMA data whose
// prefetch window (2 buffers of 64 KiB) is only one K-step deep -- while the L2 -> LDS path itself needs
// 0.92 us per 64 KiB per CU at its ceiling (~70 GB/s per CU).  This kernel keeps the ping-pong idea and
// changes the granularity:
//   * a STAGE is 64 k (A 256 x 64 B + B 256 x 64 B = 32 KiB); four stages live in LDS, so three stages of
//     LDS-DMA are in flight behind the one being read (>= 4 phases, ~1 us, of latency tolerance);
//   * a stage is consumed in TWO phases: one group of four waves issues its whole stage of MFMAs
//     (32 x v_mfma_i32_16x16x64_i8 or 8 x v_mfma_scale_f32_32x32x64_f8f6f4 = 512 matrix-pipe cycles) while
//     the other group reads its next stage's fragments (12 x ds_read_b128 / 24 x ds_read_b64_tr_b8),
//     issues its share of the LDS-DMA for stage s+4 and waits for stage s+2 -- 4 barriers per 128 k;
//   * MFMA shapes whose K divides the 64-k stage: int8 v_mfma_i32_16x16x64_i8 (32 per wave and stage),
//     fp8 v_mfma_scale_f32_32x32x64_f8f6f4 (8 per wave and stage; the 16x16 scaled form needs K = 128).
//     Shape matters beyond cycle counts: with every CU issuing back-to-back MFMAs on random operands
//     (tools/probe_mfma.hip) the chip holds 1.91 GHz on i32_16x16x64_i8 but 1.40 GHz on i32_32x32x32_i8
//     (9.15 vs 22.9 ns per MFMA and wave: 3.67 vs 2.93 POPS chip-wide), so int8 uses the 16 x 16 shape.
// Per-group program (group 1 runs one barrier behind group 0):
//     MFMA(s)  | barrier | LOAD frags(s+1); DMA stage s+4 -> buffer s & 3; vmcnt: stage s+2 landed | barrier
// Buffer safety: stage s is read by group 0 in LOAD(s-1) and by group 1 one phase later; the DMA that
// overwrites its buffer is issued in LOAD(s), at least one barrier after both.  Stage s+2 is first read
// in LOAD(s+1), after every wave's own vmcnt in LOAD(s) and the barrier(s) that follow.
#pragma once
#include <atomic>
#include "gnk_gemm_kernel.h"

typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

namespace gnk_gemm {

template <bool AK_, bool BK_, int OP_, int FA_, int FB_, int EPI_>
struct PP2Cfg {
    static constexpr bool A_KMAJOR = AK_, B_KMAJOR = BK_;
    static constexpr int OP = OP_, FA = FA_, FB = FB_, EPI = EPI_;
    static constexpr int SK = 64;                       // k per stage
    static constexpr int NST = 4;                       // stages in LDS
    static constexpr int OPER_BYTES = 256 * SK;         // one operand, one stage
    static constexpr int LDS_BYTES = 2 * NST * OPER_BYTES;   // 128 KiB
    static_assert(EPI_ != EPI_TILE, "tile-scaled epilogue uses the two-buffer kernel");
};

template <int OP> struct Acc32;
template <> struct Acc32<OP_I8> { typedef v16i type; };
template <> struct Acc32<OP_F8> { typedef v16f type; };

// This wave's two 1-KiB pieces of one operand stage (see Stager in gnk_gemm_kernel.h).
template <bool KMAJOR> struct Stager64 {
    const char *base;
    unsigned off[2];
    int64_t step;
    __device__ __forceinline__ void init(const char *ptr, int64_t ld, int64_t r0, int64_t extent, int wave, int lane) {
        base = KMAJOR ? ptr + r0 * ld : ptr + r0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = (i * 8 + wave) * 64 + lane;
            if constexpr (KMAJOR) {
                int r, c;
                gnk_k64_stage_src(p, r, c);
                const int64_t rmax = extent - 1 - r0;
                const int64_t row = r < rmax ? r : rmax;
                off[i] = (unsigned)(row * ld + 16 * c);
            } else {
                int k, c;
                gnk_mn64_stage_src(p, k, c);
                const int64_t cmax = extent - 16 - r0;
                const int64_t col = 16 * c < cmax ? 16 * c : cmax;
                off[i] = (unsigned)((int64_t)k * ld + col);
            }
        }
        step = KMAJOR ? 64 : 64 * ld;
    }
    __device__ __forceinline__ void issue(char *stage, int wave) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const GNK_GLOBAL void *)(base + off[i]), (GNK_LDS void *)(stage + (i * 8 + wave) * 1024), 16, 0, 0);
        base += step;
    }
};

__device__ __forceinline__ void pp2_phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- int8: 16 x 16 x 64 MFMA tiles; wave tile 128 x 64 = 8 x 4 tiles, fragments of 16 rows x 64 k (16 B per lane)
template <class Cfg> struct PP2I8 {
    typedef GemmCfg<256, 256, 2, 4, Cfg::A_KMAJOR, Cfg::B_KMAJOR, OP_I8, 0, 0, Cfg::EPI> ECfg;   // epilogue maps of gnk_gemm_kernel.h
    typedef v4i acc_arr[8][4];
    typedef v4i fa_arr[8];
    typedef v4i fb_arr[4];
    static __device__ __forceinline__ void zero(acc_arr &acc) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
    }
    static __device__ __forceinline__ void finish(const GemmParams &p, acc_arr &acc, char *smem, int wave, int lane, int64_t mwave, int64_t nwave,
                                                  const float *lsc, int mlw, int nlw) {
        v4f(*no_accf)[1] = nullptr;
        run_epilogue<ECfg>(p, acc, no_accf, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
    }
    template <bool KMAJOR> static __device__ __forceinline__ v4i load_frag16(const char *stage, int rb, int lane) {
        if constexpr (KMAJOR) {
            return *(const GNK_LDS v4i *)(const GNK_LDS char *)(stage + gnk_k64_frag16_offset(rb, lane));
        } else {
            const int o = gnk_mn64_frag16_offset(rb, 0, lane);
            const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(stage + o));
            const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(stage + o + 8 * 256));
            return (v4i){lo[0], lo[1], hi[0], hi[1]};
        }
    }
    static __device__ __forceinline__ void load_frags(const char *As, const char *Bs, v4i (&fa)[8], v4i (&fb)[4], int wm, int wn, int lane) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = load_frag16<Cfg::B_KMAJOR>(Bs, wn * 64 + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = load_frag16<Cfg::A_KMAJOR>(As, wm * 128 + 16 * i, lane);
    }
    static __device__ __forceinline__ void mfma_stage(v4i (&acc)[8][4], const v4i (&fa)[8], const v4i (&fb)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        __builtin_amdgcn_s_setprio(0);
    }
};

// ---- fp8: 32 x 32 x 64 scaled MFMA tiles; wave tile 128 x 64 = 4 x 2 tiles
template <class Cfg> struct PP2F8 {
    typedef typename Acc32<Cfg::OP>::type acc_t;
    typedef acc_t acc_arr[4][2];
    typedef v8i fa_arr[4];
    typedef v8i fb_arr[2];
    static __device__ __forceinline__ void zero(acc_arr &acc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    }

    // 32 bytes of one lane for the 32-row fragment starting at row / column rb of the stage image
    template <bool KMAJOR> static __device__ __forceinline__ v8i load_frag32(const char *stage, int rb, int lane) {
        if constexpr (KMAJOR) {
            const v4i lo = *(const GNK_LDS v4i *)(const GNK_LDS char *)(stage + gnk_k64_frag_offset(rb, 0, lane));
            const v4i hi = *(const GNK_LDS v4i *)(const GNK_LDS char *)(stage + gnk_k64_frag_offset(rb, 1, lane));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int o = gnk_mn64_frag_offset(rb, 0, 0, lane);   // (half, t) move the k row by 32*half + 8*t: + 256 B per row
            const v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(stage + o));
            const v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(stage + o + 8 * 256));
            const v2i c = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(stage + o + 32 * 256));
            const v2i d = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(stage + o + 40 * 256));
            return (v8i){a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
        }
    }

    static __device__ __forceinline__ void load_frags(const char *As, const char *Bs, v8i (&fa)[4], v8i (&fb)[2], int wm, int wn, int lane) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = load_frag32<Cfg::B_KMAJOR>(Bs, wn * 64 + 32 * j, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = load_frag32<Cfg::A_KMAJOR>(As, wm * 128 + 32 * i, lane);
    }

    // one stage of MFMAs: acc[i][j] (32 m x 32 n, held as D[n][m]) += B-frag(j) x A-frag(i)
    static __device__ __forceinline__ void mfma_stage(acc_t (&acc)[4][2], const v8i (&fa)[4], const v8i (&fb)[2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j], fa[i], acc[i][j], Cfg::FB, Cfg::FA, 0, 0x7f7f7f7f, 0,
                                                                             0x7f7f7f7f);   // E8M0 127 = 2^0
        // pin the cluster to this phase (register-only ops are otherwise free to sink past the barriers)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
        __builtin_amdgcn_s_setprio(0);
    }

    // ------------------------------------------------------------------ epilogue (32 x 32 tile maps)
    // lane: m = mwave + 32 i + (lane & 31);  n = nwave + 32 j + 8 g + 4 h + r  with reg = 4 g + r, h = lane >> 5
    template <int ORDER> static __device__ __forceinline__ float value(const GemmParams &p, const acc_t &a, int reg, float rs, float cs) {
        float x;
        if constexpr (Cfg::OP == OP_I8) x = (float)a[reg];   // v_cvt_f32_i32: RNE
        else x = a[reg];
        if constexpr (Cfg::EPI == EPI_PLAIN) return x;
        if constexpr (ORDER == 0) return __fmul_rn(__fmul_rn(x, rs), cs);
        static_assert(ORDER == 0 || ORDER == 1, "multiply order is a compile-time choice (EpiScales::value)");
        return __fmul_rn(__fmul_rn(x, cs), rs);
    }

    template <int OUT, int MODE, int ORDER = 0>   // ORDER: multiply order resolved at compile time (see EpiScales::value)   // MODE 0: LDS-staged 16-byte stores (2-byte outputs); 1: direct vector; 2: direct scalar
    static __device__ __forceinline__ void epilogue(const GemmParams &p, acc_t (&acc)[4][2], char *smem, int wave, int lane, int64_t mwave, int64_t nwave,
                                                    const float *lsc, int mlw, int nlw) {
        float rs[4], cs[2][4][4];
        if constexpr (Cfg::EPI == EPI_ROWCOL) {   // scales were prefetched into LDS at kernel start (ScalePrefetch): rows, then columns
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] = *(const GNK_LDS float *)(lsc + mlw + 32 * i + (lane & 31));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f c = *(const GNK_LDS v4f *)(lsc + 256 + nlw + 32 * j + 8 * g + 4 * (lane >> 5));
                    cs[j][g][0] = c[0]; cs[j][g][1] = c[1]; cs[j][g][2] = c[2]; cs[j][g][3] = c[3];
                }
        }
        const int ml = lane & 31, h = lane >> 5;
        char *const region = smem + wave * (128 * 128);   // this wave's 128 rows x 128 B result slice
        // store side of the staged form: see epilogue_lds (one 64-bit address per lane, wave-uniform step)
        const int srow = lane >> 3, sc8 = lane & 7;
        const int64_t rows_left64 = p.M - mwave - srow;
        const int rows_left = rows_left64 > 128 ? 128 : (int)rows_left64;
        const bool col_ok = nwave + 8 * sc8 < p.N;
        unsigned short *const cbase = (unsigned short *)p.C + (mwave + srow) * p.N + nwave + 8 * sc8;
        const int64_t cstep = 8 * p.N;
        v4i dq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (OUT == GNK_I32) v[r] = __int_as_float(acc[i][j][4 * g + r]);
                        else v[r] = value<ORDER>(p, acc[i][j], 4 * g + r, rs[i], cs[j][g][r]);
                    }
                    if constexpr (MODE == 0) {
                        v2i o;
                        if constexpr (OUT == GNK_BF16) {
                            o[0] = (int)gnk_pack2_bf16(v[0], v[1]);
                            o[1] = (int)gnk_pack2_bf16(v[2], v[3]);
                        } else {
                            o[0] = (int)gnk_pack2_f16(v[0], v[1]);
                            o[1] = (int)gnk_pack2_f16(v[2], v[3]);
                        }
                        const int row = 32 * i + ml;
                        *(GNK_LDS v2i *)(GNK_LDS char *)(region + row * 128 + (((4 * j + g) ^ (row & 7)) << 4) + 8 * h) = o;
                    } else {
                        store4<OUT, MODE == 1>(p.C, mwave + 32 * i + ml, nwave + 32 * j + 8 * g + 4 * h, p.M, p.N, v[0], v[1], v[2], v[3]);
                    }
                }
            if constexpr (MODE == 0) {   // rows 32 i .. 32 i + 31 are complete: read them back now, store them behind the NEXT row block's
                                         // conversion (epilogue_lds: one wave per SIMD has nothing else to hide the LDS round trip with)
                if (i > 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int it = 4 * (i - 1) + u;
                        if (col_ok && it * 8 < rows_left) gnk_store16_out(cbase + it * cstep, dq[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = (4 * i + u) * 8 + srow;
                    dq[u] = *(const GNK_LDS v4i *)(const GNK_LDS char *)(region + row * 128 + ((sc8 ^ (row & 7)) << 4));
                }
            }
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = 12 + u;
                if (col_ok && it * 8 < rows_left) gnk_store16_out(cbase + it * cstep, dq[u]);
            }
        }
    }
    static __device__ __forceinline__ void finish(const GemmParams &p, acc_arr &acc, char *smem, int wave, int lane, int64_t mwave, int64_t nwave,
                                                  const float *lsc, int mlw, int nlw) {
        const bool aligned16 = ((uintptr_t)p.C & 15) == 0;
        const bool vec_ok = (p.N % 4 == 0) && aligned16, lds_ok = (p.N % 8 == 0) && aligned16;
        if constexpr (Cfg::EPI == EPI_PLAIN && Cfg::OP == OP_I8) {
            if (vec_ok) epilogue<GNK_I32, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
            else epilogue<GNK_I32, 2>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
        } else if (p.out_dtype == GNK_BF16) {
            if (lds_ok) {
                if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_BF16, 0, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
                else epilogue<GNK_BF16, 0, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
            }
            else if (vec_ok) { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_BF16, 1, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); else epilogue<GNK_BF16, 1, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); }
            else { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_BF16, 2, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); else epilogue<GNK_BF16, 2, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); }
        } else if (p.out_dtype == GNK_F16) {
            if (lds_ok) {
                if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_F16, 0, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
                else epilogue<GNK_F16, 0, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
            }
            else if (vec_ok) { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_F16, 1, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); else epilogue<GNK_F16, 1, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); }
            else { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_F16, 2, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); else epilogue<GNK_F16, 2, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); }
        } else {
            if (vec_ok) { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_F32, 1, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); else epilogue<GNK_F32, 1, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); }
            else { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<GNK_F32, 2, 0>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); else epilogue<GNK_F32, 2, 1>(p, acc, smem, wave, lane, mwave, nwave, lsc, mlw, nlw); }
        }
    }
};

template <class Cfg, int OP> struct PP2Select;
template <class Cfg> struct PP2Select<Cfg, OP_I8> { typedef PP2I8<Cfg> type; };
template <class Cfg> struct PP2Select<Cfg, OP_F8> { typedef PP2F8<Cfg> type; };

template <class Cfg>
__global__ void __launch_bounds__(512) gemm_pp2_kernel(const GemmParams p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename PP2Select<Cfg, Cfg::OP>::type K;

    // split-K (plain int32 output only), as in gemm_kernel: split s = blockIdx.y covers k in [s * kchunk, min(K, (s + 1) * kchunk)) and owns
    // slab s of the workspace; integer partial sums add exactly.  Wave-uniform scalar arithmetic.
    GemmParams p = p_in;
    if constexpr (Cfg::EPI == EPI_PLAIN && Cfg::OP == OP_I8) {
        if (p.kchunk > 0) {
            // split 0 may be longer than the others (kchunk0): when the tiles do not fill the chip, the first splits of all tiles occupy
            // their CUs for the whole launch and the remaining CUs work through the short second splits in rounds (gnk_api.hip:plan_gemm)
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

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

    ScalePrefetch<256, 256, 512, Cfg::EPI> spf;
    spf.issue(p, m0, n0);

    Stager64<Cfg::A_KMAJOR> sa;
    Stager64<Cfg::B_KMAJOR> sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);

    typename K::acc_arr acc;
    K::zero(acc);
    typename K::fa_arr fa;
    typename K::fb_arr fb;

    const int ns = (int)(p.K / Cfg::SK);
    char *const Abuf = smem, *const Bbuf = smem + Cfg::NST * Cfg::OPER_BYTES;

    // prologue: stages 0..3 in flight, wait for stage 0 only
    const int npro = ns < Cfg::NST ? ns : Cfg::NST;
    for (int s = 0; s < npro; ++s) {
        sa.issue(Abuf + s * Cfg::OPER_BYTES, wave);
        sb.issue(Bbuf + s * Cfg::OPER_BYTES, wave);
    }
    if (npro == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (npro == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (npro == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp2_phase_barrier();
    K::load_frags(Abuf, Bbuf, fa, fb, wm, wn, lane);
    // stage 1 must have landed before the first LOAD phase reads it
    if (npro == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (npro == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pp2_phase_barrier();
    if (wm == 1) pp2_phase_barrier();   // stagger: group 1 runs one phase behind group 0

    for (int s = 0; s < ns; ++s) {
        // ---- MFMA phase
        K::mfma_stage(acc, fa, fb);
        pp2_phase_barrier();
        // ---- LOAD phase
        if (s + 1 < ns) {
            const int b = (s + 1) & 3;
            K::load_frags(Abuf + b * Cfg::OPER_BYTES, Bbuf + b * Cfg::OPER_BYTES, fa, fb, wm, wn, lane);
        }
        if (s + 4 < ns) {
            const int b = s & 3;
            sa.issue(Abuf + b * Cfg::OPER_BYTES, wave);
            sb.issue(Bbuf + b * Cfg::OPER_BYTES, wave);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stages s+3, s+4 may stay in flight; s+2 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: drain
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp2_phase_barrier();
    }
    if (wm == 0) pp2_phase_barrier();   // pay the stagger back so barrier counts match

    // all LDS stage reads retired at the last barrier; each wave stages its result through its own LDS slice
    const int64_t mwave = m0 + wm * 128, nwave = n0 + wn * 64;
    spf.park(p, smem + Cfg::LDS_BYTES);
    __syncthreads();   // scales parked in LDS are visible to every wave
    K::finish(p, acc, smem, wave, lane, mwave, nwave, Cfg::EPI == EPI_ROWCOL ? (const float *)(smem + Cfg::LDS_BYTES) : nullptr, wm * 128, wn * 64);
}

template <class Cfg> int launch_pp2(const GemmParams &p0, hipStream_t st) {
    constexpr int LDS_TOTAL = Cfg::LDS_BYTES + ScalePrefetch<256, 256, 512, Cfg::EPI>::BYTES;
    GemmParams p = p0;
    p.grid_m = (int)((p.M + 255) / 256);
    p.grid_n = (int)((p.N + 255) / 256);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "gemm: no current HIP device");
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void *)gemm_pp2_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return gnk_fail(GNK_ERR_LAUNCH, "gemm: cannot reserve %d bytes of LDS", LDS_TOTAL);
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int64_t c0 = p.kchunk0 > 0 ? p.kchunk0 : p.kchunk;
    const unsigned splits = p.kchunk > 0 ? 1u + (unsigned)((p.K - c0 + p.kchunk - 1) / p.kchunk) : 1u;
    hipLaunchKernelGGL((gemm_pp2_kernel<Cfg>), dim3((unsigned)(p.grid_m * p.grid_n), splits), dim3(512), LDS_TOTAL, st, p);
    return gnk_check_launch("gemm (4-stage ping-pong)");
}

}  // namespace gnk_gemm
