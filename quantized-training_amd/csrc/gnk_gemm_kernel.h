// gnk_gemm_kernel.h -- LDS-staged MFMA GEMM for 8-bit operands on gfx950 (CDNA4).
//
//   C(M x N) = epilogue( A(M x K) @ B(K x N) ),   A, B: int8 or OCP fp8 (1 byte / element)
//
// One kernel template covers the reference's three GEMM families and all operand layouts:
//   EPI_PLAIN : int32 (int8) or fp32->out (fp8) result      triton_mm.py:53-143, cutlass fp8_mm
//   EPI_ROWCOL: (f32(acc) * row[m]) * col[n] -> bf16/f16/f32 triton_mm.py:146-220,382-407
//   EPI_TILE  : per-128-K-block rescale, fp32 accumulate     triton_mm.py:223-342,410-441
//
// Design (MI355X-first, not a translation of the Triton kernels):
//   * operands are staged global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two
//     LDS buffers, the next K-step's loads in flight while the current one feeds the MFMAs;
//   * a K-contiguous operand is kept as [rows][128 B] and read with ds_read_b128; an M/N-contiguous
//     operand (the "column-major" layouts A@B and A.T@B of benchmark_mm.py:33-34) is kept exactly
//     as it lies in memory, [128 k][rows], and transposed on the LDS->register hop with
//     ds_read_b64_tr_b8, so global loads stay 16-byte coalesced and no transpose is materialised;
//   * both images are XOR-swizzled (gnk_lds_layout.h) so fragment reads are bank-conflict free;
//   * MFMA: v_mfma_i32_16x16x64_i8, or v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales
//     (the only fp8 form that runs at 2x the bf16 rate on CDNA4);
//   * the MFMA "A" operand is fed from matrix B and vice versa, so each lane ends up with 4
//     consecutive n for one m: the epilogue packs them into one 8-byte (bf16) / 16-byte store;
//   * workgroups are remapped so that the 8 XCDs (private L2s) each own a compact patch of tiles.
#pragma once
#include <atomic>
#include "gnk_common.h"
#include "gnk_lds_layout.h"

enum { EPI_PLAIN = 0, EPI_ROWCOL = 1, EPI_TILE = 2 };
enum { OP_I8 = 0, OP_F8 = 1 };

struct GemmParams {
    const char *A;
    const char *B;
    void *C;
    const void *s0;   // row scale (M) or tile scale A
    const void *s1;   // col scale (N or 1) or tile scale B
    int64_t M, N, K;
    int64_t lda;      // elements between consecutive m (A K-major) or consecutive k (A M-major)
    int64_t ldb;      // elements between consecutive n (B K-major) or consecutive k (B N-major)
    int64_t ssam, ssak, ssbk, ssbn;   // tile-scale strides
    int64_t QM, QN;                   // tile-scale block sizes along M and N (QK is the kernel's Cfg::KB)
    int out_dtype, scale_dtype, col_scalar, order;
    int grid_m, grid_n;
    int kgroup;       // tile-scaled with QK = kgroup * 128 (kgroup >= 1; 1 for the 64 / 32-k block kernels): K-steps per scale block
    int64_t kchunk;   // split-K: k elements per split (gridDim.y splits, each writing its own (M, N) int32 slab at C); 0 = no split
    int64_t kchunk0;  // 256 x 256 kernel only: k elements of split 0 when it differs from the others (uneven split); 0 = kchunk
    int fused_rescale;   // tile-scaled: 1 = acc = fma(f32(part), sA * sB, acc) (tolerance mode), 0 = the reference interpreter's three roundings
    int debug;        // timing-ablation flags (gnk_set_debug_flags): 1 = no LDS-DMA in the main loop, 2 = no MFMA, 4 = no LDS reads
};

// KB_: k per scale block of the tile-scaled epilogue (128: one block per K-step; 64 / 32: Jetfire-style blocks, 2 / 4 per step;
// 256 stands for ANY multiple of 128 above 128: GemmParams::kgroup K-steps per block, a run-time number)
// FUSED_: tile-scaled rescale as one fma per element with s = sA * sB (GemmParams::fused_rescale; its own instances: as a run-time branch
// beside the three-rounding form the 256-register kernels spilled 14-20 registers)
// PERCOL_: tile-scaled with a scale block width QN that is NOT a multiple of 4 (per-column scales, QN = 1, 2, 3 ...): a lane's four consecutive
// columns may sit in different blocks, so the B-side scales go through the LDS panel expanded per COLUMN and every element takes its own
// (128-k block instances only; before round 3 these shapes fell to the generic kernel, 44-49 TOP/s)
template <int BM_, int BN_, int WM_, int WN_, bool AK_, bool BK_, int OP_, int FA_, int FB_, int EPI_, int KB_ = GNK_BK, bool FUSED_ = false, bool PERCOL_ = false>
struct GemmCfg {
    static constexpr bool FUSED = FUSED_, PERCOL = PERCOL_;
    static_assert(!PERCOL_ || (EPI_ == 2 && KB_ == GNK_BK), "per-column B scales: tile-scaled epilogue with 128-k blocks");
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, KB = KB_;
    static_assert(KB_ == 128 || ((KB_ == 64 || KB_ == 32 || KB_ == 256) && EPI_ == 2), "scale blocks other than 128 k exist for the tile-scaled epilogue only");
    static constexpr bool A_KMAJOR = AK_, B_KMAJOR = BK_;
    static constexpr int OP = OP_, FA = FA_, FB = FB_, EPI = EPI_;
    static constexpr int NW = WM * WN, NT = NW * 64;
    static constexpr int WTM = BM / WM, WTN = BN / WN;    // wave tile
    static constexpr int MT = WTM / 16, NTL = WTN / 16;   // 16x16 MFMA tiles per wave
    static constexpr int A_BYTES = BM * GNK_BK, B_BYTES = BN * GNK_BK;
    static constexpr int LDS_BYTES = 2 * (A_BYTES + B_BYTES);
    static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "wave tile must be a multiple of 16");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "1-KiB staging pieces must divide over the waves");
};

namespace gnk_gemm {

// ------------------------------------------------------------------------------ tile mapping
// XCD-aware: blocks b, b+8, b+16 ... share an XCD (observed round-robin placement; speed only).
// Give every XCD a contiguous run of tile ids, then walk tiles in GROUP_M-tall column strips so
// the run is a compact patch (e.g. 4 x 8 tiles at 4096^2 / 256^2 tiles).
__device__ __forceinline__ void tile_of_block(int grid_m, int grid_n, int &tm, int &tn) {
    const int nwg = grid_m * grid_n;
    const int b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective
    constexpr int GROUP_M = 4;
    const int width = GROUP_M * grid_n;
    const int group = wg / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(grid_m - first_m, GROUP_M);
    const int in = wg - group * width;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

// ------------------------------------------------------------------------------ staging
// This wave's 1-KiB pieces of one operand tile: a wave-uniform base pointer (SGPRs, advanced once per
// K-step) plus one 32-bit byte offset per piece and lane (constant over the whole K loop), so the
// LDS-DMA loads use the saddr + voffset form and cost no per-step address VALU.
template <bool KMAJOR, int ROWS, int NW>
struct Stager {
    static constexpr int NP = ROWS / 8 / NW;   // pieces per wave
    const char *base;                          // tile origin at the current K-step (wave-uniform)
    unsigned off[NP];                          // per-lane byte offset of each piece from `base`
    int64_t step;                              // bytes to advance per K-step

    // ptr: operand pointer; ld: leading dimension; r0: first row (m0 / n0) of the tile; extent: M or N.
    // Rows / column chunks past the extent are clamped so that every lane reads valid memory; they
    // only feed output elements that the epilogue masks.  The host guarantees 256 * ld < 2^31.
    __device__ __forceinline__ void init(const char *ptr, int64_t ld, int64_t r0, int64_t extent, int wave, int lane) {
        base = KMAJOR ? ptr + r0 * ld : ptr + r0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pc = i * NW + wave;
            const int p = pc * 64 + lane;
            if constexpr (KMAJOR) {
                int r, c;
                gnk_kmaj_stage_src(p, r, c);
                const int64_t rmax = extent - 1 - r0;
                const int64_t row = r < rmax ? r : rmax;
                off[i] = (unsigned)(row * ld + 16 * c);
            } else {
                int k, c;
                gnk_mnmaj_stage_src<ROWS>(p, k, c);
                const int64_t cmax = extent - 16 - r0;
                const int64_t col = 16 * c < cmax ? 16 * c : cmax;
                off[i] = (unsigned)((int64_t)k * ld + col);
            }
        }
        step = KMAJOR ? (int64_t)GNK_BK : (int64_t)GNK_BK * ld;
    }

    // issue this wave's pieces of the tile at `base` into the LDS tile, then move on to the next K-step
    __device__ __forceinline__ void issue(char *tile, int wave) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pc = i * NW + wave;
            __builtin_amdgcn_global_load_lds((const GNK_GLOBAL void *)(base + off[i]), (GNK_LDS void *)(tile + pc * 1024), 16, 0, 0);
        }
        base += step;
    }

    // The LAST K-step when K % 128 != 0: only k < kvalid of the tile exist.  Pieces that lie wholly beyond them are fetched from the
    // tile's first chunk / k-row instead (valid memory; a K-major chunk that straddles kvalid stays inside the row pitch, which the
    // host requires to be >= K rounded up to 16): what they bring is garbage that zero_tail() clears before any fragment is read.
    __device__ __forceinline__ void issue_tail(char *tile, int64_t ld, int64_t r0, int64_t extent, int wave, int lane, int kvalid) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pc = i * NW + wave;
            const int p = pc * 64 + lane;
            unsigned o = off[i];
            if constexpr (KMAJOR) {
                int r, c;
                gnk_kmaj_stage_src(p, r, c);
                if (16 * c >= kvalid) o -= 16u * (unsigned)c;                       // chunk 0 of the same row
            } else {
                int k, c;
                gnk_mnmaj_stage_src<ROWS>(p, k, c);
                if (k >= kvalid) o -= (unsigned)((int64_t)k * ld);                  // k-row 0, same columns
            }
            __builtin_amdgcn_global_load_lds((const GNK_GLOBAL void *)(base + o), (GNK_LDS void *)(tile + pc * 1024), 16, 0, 0);
        }
        base += step;
    }
};

// Clear k >= kvalid of a staged tile (the last K-step of a K that is not a multiple of 128): zero bytes are 0 in int8 and +0 in both
// fp8 formats, so the tail adds nothing to any dot product.  Called by every thread after the tile has landed (vmcnt(0) + barrier);
// the caller puts a barrier behind it.
template <bool KMAJOR, int ROWS, int NT>
__device__ __forceinline__ void zero_tail(char *tile, int kvalid) {
    if constexpr (KMAJOR) {
        const int d0 = kvalid >> 2, nd = 32 - d0;                                    // dwords d0 .. 31 of every row hold k >= 4 d0
        for (int e = threadIdx.x; e < ROWS * nd; e += NT) {
            const int r = e / nd, d = d0 + e % nd;
            GNK_LDS unsigned *w = (GNK_LDS unsigned *)(GNK_LDS char *)(tile + gnk_kmaj_offset(r, d >> 2) + 4 * (d & 3));
            const int keep = kvalid - 4 * d;                                         // bytes of this dword that are real (<= 3 here, or <= 0)
            *w = keep > 0 ? (*w & ((1u << (8 * keep)) - 1u)) : 0u;
        }
    } else {
        // k-rows are ROWS contiguous bytes (the swizzle permutes chunks inside a row): one linear fill
        for (int o = kvalid * ROWS + 16 * (int)threadIdx.x; o < 128 * ROWS; o += 16 * NT)
            *(GNK_LDS v4i *)(GNK_LDS char *)(tile + o) = (v4i){0, 0, 0, 0};
    }
}

// ------------------------------------------------------------------------------ fragment reads
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ v4i load_frag(const char *tile, int rb, int ks, int lane) {
    if constexpr (KMAJOR) {
        return *(const GNK_LDS v4i *)(const GNK_LDS char *)(tile + gnk_kmaj_frag_offset(rb, ks, lane));
    } else {
        const int off = gnk_mnmaj_frag_offset<ROWS>(rb, ks, lane);
        v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(tile + off));
        v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(tile + off + 8 * ROWS));
        return (v4i){lo[0], lo[1], hi[0], hi[1]};
    }
}

template <int OP> struct AccT;
template <> struct AccT<OP_I8> { typedef v4i type; };
template <> struct AccT<OP_F8> { typedef v4f type; };

// one K-step (128 k) of MFMAs for this wave.  FRESH: start from C = 0 (inline constant: the tile-scaled
// kernel rescales every K-block's partial sum separately and needs no accumulator re-zeroing).
template <class Cfg, bool FRESH = false>
__device__ __forceinline__ void compute_step(const char *As, const char *Bs, typename AccT<Cfg::OP>::type (&acc)[Cfg::MT][Cfg::NTL],
                                             int wm, int wn, int lane) {
    constexpr int MT = Cfg::MT, NTL = Cfg::NTL;
    if constexpr (Cfg::OP == OP_I8) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i fa[MT], fb[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) fb[j] = load_frag<Cfg::B_KMAJOR, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, ks, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = load_frag<Cfg::A_KMAJOR, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, ks, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[j], fa[i], (FRESH && ks == 0) ? (v4i){0, 0, 0, 0} : acc[i][j], 0, 0, 0);
        }
    } else {
        // fp8: one 16x16x128 MFMA per tile pair; lane slot (g, byte) <-> k = 64*(byte/16) + 16g + byte%16
        // for BOTH operands (any consistent k permutation is a valid dot product).
        v8i fb[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            v4i lo = load_frag<Cfg::B_KMAJOR, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, 0, lane);
            v4i hi = load_frag<Cfg::B_KMAJOR, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, 1, lane);
            fb[j] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            v4i lo = load_frag<Cfg::A_KMAJOR, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, 0, lane);
            v4i hi = load_frag<Cfg::A_KMAJOR, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, 1, lane);
            v8i fa = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa, FRESH ? (v4f){0.f, 0.f, 0.f, 0.f} : acc[i][j], Cfg::FB, Cfg::FA, 0,
                                                                             0x7f7f7f7f, 0, 0x7f7f7f7f);   // E8M0 127 = 2^0
        }
    }
}

// ---- 32-k sub-blocks (tile-scaled with 64 / 32-k scale blocks: the partial sum must stop at a block edge, which the 64-k and
// 128-k MFMAs cannot).  Lane l = 16 g + r supplies row r and k = 32 s + 8 g .. + 7 of sub-block s (0..3) of the 128-k tile:
// K-major: half (g & 1) of 16-byte chunk 2 s + (g >> 1); MN-major: one ds_read_b64_tr_b8 over k rows 32 s + 8 g .. + 7.
template <bool KMAJOR, int ROWS>
__device__ __forceinline__ long load_frag8(const char *tile, int rb, int s, int lane) {
    v2i v;
    if constexpr (KMAJOR) {
        // == gnk_kmaj_offset(rb + r, 2 s + (g >> 1)) + 8 (g & 1), written so that ONE per-lane term depends on s: rb is a multiple of 16, so
        // the row's swizzle ((rb + r) >> 1) & 7 is (r >> 1) & 7 for every fragment, and (2 s + gh) ^ sw == (2 s) ^ (gh ^ sw) (gh is bit 0).
        // Without this the compiler kept one address register per (fragment, sub-block): 32 per operand.
        const int g = lane >> 4, r = lane & 15;
        const int u16 = (((g >> 1) ^ (r >> 1)) & 7) << 4;
        v = *(const GNK_LDS v2i *)(const GNK_LDS char *)(tile + rb * 128 + (r * 128 + 8 * (g & 1)) + ((32 * s) ^ u16));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 1, hb = lane & 1;
        v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((GNK_LDS v2i *)(tile + gnk_mnmaj_offset<ROWS>(32 * s + 8 * g + q, rb >> 4) + 8 * hb));
    }
    return __builtin_bit_cast(long, v);
}

// one 32-k sub-block of MFMAs for this wave: v_mfma_i32_16x16x32_i8 / v_mfma_f32_16x16x32_{fp8,bf8}_{fp8,bf8}
template <class Cfg>
__device__ __forceinline__ void compute_sub32(const char *As, const char *Bs, typename AccT<Cfg::OP>::type (&acc)[Cfg::MT][Cfg::NTL],
                                              int wm, int wn, int lane, int s, bool fresh) {
    constexpr int MT = Cfg::MT, NTL = Cfg::NTL;
    long fa[MT], fb[NTL];
    asm volatile("" ::: "memory");   // compiler-only fence: without it the fragment reads of all sub-blocks were hoisted to the top (spills)
#pragma unroll
    for (int j = 0; j < NTL; ++j) fb[j] = load_frag8<Cfg::B_KMAJOR, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, s, lane);
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = load_frag8<Cfg::A_KMAJOR, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, s, lane);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            if constexpr (Cfg::OP == OP_I8) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(fb[j], fa[i], fresh ? (v4i){0, 0, 0, 0} : acc[i][j], 0, 0, 0);
            } else {   // the first MFMA operand is the B fragment: its format is FB (0 = e4m3 "fp8", 1 = e5m2 "bf8")
                const v4f c = fresh ? (v4f){0.f, 0.f, 0.f, 0.f} : acc[i][j];
                if constexpr (Cfg::FB == 0 && Cfg::FA == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j], fa[i], c, 0, 0, 0);
                else if constexpr (Cfg::FB == 0 && Cfg::FA == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[j], fa[i], c, 0, 0, 0);
                else if constexpr (Cfg::FB == 1 && Cfg::FA == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(fb[j], fa[i], c, 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(fb[j], fa[i], c, 0, 0, 0);
            }
        }
}

// ------------------------------------------------------------------------------ epilogue stores
// OUT is a compile-time dtype and VEC says whether the 4 consecutive n of a lane can go out as one
// 8/16-byte store (N % 4 == 0 and C 16-byte aligned); both are hoisted out of the tile loops so the
// epilogue is straight-line code.
template <int OUT, bool VEC>
__device__ __forceinline__ void store4(void *C, int64_t m, int64_t n, int64_t M, int64_t N, float v0, float v1, float v2, float v3) {
    if (m >= M || n >= N) return;
    const int64_t idx = m * N + n;
    if constexpr (VEC) {
        if constexpr (OUT == GNK_BF16) {
            v2i o = {(int)gnk_pack2_bf16(v0, v1),
                     (int)gnk_pack2_bf16(v2, v3)};
            *(v2i *)((unsigned short *)C + idx) = o;
        } else if constexpr (OUT == GNK_F16) {
            v2i o = {(int)gnk_pack2_f16(v0, v1),
                     (int)gnk_pack2_f16(v2, v3)};
            *(v2i *)((unsigned short *)C + idx) = o;
        } else if constexpr (OUT == GNK_I32) {
            *(v4i *)((int *)C + idx) = (v4i){__float_as_int(v0), __float_as_int(v1), __float_as_int(v2), __float_as_int(v3)};
        } else {
            *(v4f *)((float *)C + idx) = (v4f){v0, v1, v2, v3};
        }
    } else {
        const float v[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < N) {
                if constexpr (OUT == GNK_BF16) ((unsigned short *)C)[idx + r] = gnk_f32_to_bf16(v[r]);
                else if constexpr (OUT == GNK_F16) ((unsigned short *)C)[idx + r] = gnk_f32_to_f16(v[r]);
                else if constexpr (OUT == GNK_I32) ((int *)C)[idx + r] = __float_as_int(v[r]);
                else ((float *)C)[idx + r] = v[r];
            }
    }
}

// Final fp32 value of accumulator element (i, j, r) of this lane after the fused scale epilogue.
// Row / column scales of the block tile, fetched at kernel start (one element per thread: BM + BN == threads) and parked
// in LDS behind the operand buffers, so that the epilogue does not begin with a round trip to HBM (~1.5 us exposed
// per launch: 2.5 % of a 4096^3 product, 6 % of the K = 1024 products of a Llama-470m step).
template <int BM, int BN, int NT, int EPI> struct ScalePrefetch {
    static constexpr int BYTES = EPI == EPI_ROWCOL ? (BM + BN) * 4 : 0;
    static_assert(BM + BN <= NT, "at most one scale per thread");
    unsigned raw;
    __device__ __forceinline__ void issue(const GemmParams &p, int64_t m0, int64_t n0) {
        if constexpr (EPI == EPI_ROWCOL) {
            const int t = (int)threadIdx.x < BM + BN ? (int)threadIdx.x : BM + BN - 1;   // 64 x 128 tiles: the last 64 threads repeat a load
            const bool row = t < BM;
            int64_t idx = row ? m0 + t : n0 + (t - BM);
            const int64_t last = row ? p.M - 1 : p.N - 1;
            idx = idx < last ? idx : last;
            if (!row && p.col_scalar) idx = 0;
            raw = gnk_load_raw_rt(row ? p.s0 : p.s1, p.scale_dtype, idx);
        }
    }
    // call once the load has certainly landed (after the K loop) and before a workgroup barrier that precedes the epilogue
    __device__ __forceinline__ void park(const GemmParams &p, char *lds) const {
        if constexpr (EPI == EPI_ROWCOL) {
            const float v = gnk_raw_to_f32_rt(raw, p.scale_dtype);
            if ((int)threadIdx.x < BM + BN) *(GNK_LDS float *)(GNK_LDS char *)(lds + 4 * threadIdx.x) = v;
        }
    }
};

template <class Cfg> struct EpiScales {
    float rs[Cfg::MT];
    float cs[Cfg::NTL][4];
    // The scale dtype is switched ONCE, outside the loops, so that each branch is a run of independent
    // loads the hardware can keep in flight together (a per-load dtype branch serialised all 24 loads:
    // +9 us on a 4096^2 output).
    template <int SDT> __device__ __forceinline__ void load_t(const GemmParams &p, int64_t mrow, int64_t ncol) {
#pragma unroll
        for (int i = 0; i < Cfg::MT; ++i) {
            const int64_t m = mrow + 16 * i;
            rs[i] = gnk_load_scalar<SDT>(p.s0, m < p.M ? m : p.M - 1);
        }
#pragma unroll
        for (int j = 0; j < Cfg::NTL; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = ncol + 16 * j + r;
                cs[j][r] = gnk_load_scalar<SDT>(p.s1, p.col_scalar ? 0 : (n < p.N ? n : p.N - 1));
            }
    }
    // lds != nullptr: the block's scales were prefetched into LDS (ScalePrefetch) as fp32, rows first; mloc / nloc are
    // this lane's first row / column inside the block tile.
    __device__ __forceinline__ void load(const GemmParams &p, int64_t mrow, int64_t ncol, const float *lds = nullptr, int mloc = 0, int nloc = 0) {
        if constexpr (Cfg::EPI == EPI_ROWCOL) {
            if (lds) {
#pragma unroll
                for (int i = 0; i < Cfg::MT; ++i) rs[i] = *(const GNK_LDS float *)(lds + mloc + 16 * i);
#pragma unroll
                for (int j = 0; j < Cfg::NTL; ++j) {
                    const v4f c = *(const GNK_LDS v4f *)(lds + Cfg::BM + nloc + 16 * j);
                    cs[j][0] = c[0]; cs[j][1] = c[1]; cs[j][2] = c[2]; cs[j][3] = c[3];
                }
            } else if (p.scale_dtype == GNK_BF16) load_t<GNK_BF16>(p, mrow, ncol);
            else if (p.scale_dtype == GNK_F16) load_t<GNK_F16>(p, mrow, ncol);
            else load_t<GNK_F32>(p, mrow, ncol);
        }
    }
    // ORDER 0: (acc * row) * col (triton_mm.py:207-212); 1: (acc * col) * row (the CUTLASS-named ops).  Every epilogue resolves p.order
    // ONCE, at its entry, into one of two instances: a per-element `p.order == 0 ? ... : ...` made the compiler select between the
    // ADDRESSES of rs[] / cs[] elements, which kept both arrays in scratch memory -- 84-100 bytes of private segment per lane in every
    // row/col-scaled kernel of rounds 1-2, and with it the scratch set-up of every such dispatch.
    template <int ORDER, class AccT_>
    __device__ __forceinline__ float value(const GemmParams &p, const AccT_ &acc, const v4f *accf, int i, int j, int r) const {
        if constexpr (Cfg::EPI == EPI_TILE) return (*accf)[r];
        float a;
        if constexpr (Cfg::OP == OP_I8) a = (float)acc[r];   // v_cvt_f32_i32: RNE
        else a = acc[r];
        if constexpr (Cfg::EPI == EPI_PLAIN) return a;
        if constexpr (ORDER == 0) return __fmul_rn(__fmul_rn(a, rs[i]), cs[j][r]);
        static_assert(ORDER == 0 || ORDER == 1, "multiply order is a compile-time choice");
        return __fmul_rn(__fmul_rn(a, cs[j][r]), rs[i]);
    }
    // Elements r, r + 1 of one lane's tile row at once: the two scale multiplies as v_pk_mul_f32 (each half rounded like v_mul_f32, so
    // the bits do not change; the staged epilogues are VALU-issue bound with one or two waves per SIMD: 5.1 us of a 49 us 4096^3 launch).
    template <int ORDER, class AccT_>
    __device__ __forceinline__ v2f value2(const GemmParams &p, const AccT_ &acc, const v4f *accf, int i, int j, int r) const {
        if constexpr (Cfg::EPI == EPI_TILE) return (v2f){(*accf)[r], (*accf)[r + 1]};
        v2f a;
        if constexpr (Cfg::OP == OP_I8) a = (v2f){(float)acc[r], (float)acc[r + 1]};
        else a = (v2f){acc[r], acc[r + 1]};
        if constexpr (Cfg::EPI == EPI_PLAIN) return a;
        const v2f r2 = {rs[i], rs[i]}, c2 = {cs[j][r], cs[j][r + 1]};
        if constexpr (ORDER == 0) return (a * r2) * c2;
        return (a * c2) * r2;
    }
};

// Final values of one lane -> memory.  For OUT == GNK_I32 the floats carry int32 bit patterns.
template <class Cfg, int OUT, bool VEC, int ORDER = 0>
__device__ __forceinline__ void epilogue(const GemmParams &p, typename AccT<Cfg::OP>::type (&acc)[Cfg::MT][Cfg::NTL],
                                         v4f (*accf)[Cfg::EPI == EPI_TILE ? Cfg::NTL : 1], int64_t mrow, int64_t ncol,
                                         const float *lsc = nullptr, int mloc = 0, int nloc = 0) {
    constexpr int MT = Cfg::MT, NTL = Cfg::NTL;
    if constexpr (Cfg::EPI == EPI_PLAIN) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                if constexpr (Cfg::OP == OP_I8)
                    store4<OUT, VEC>(p.C, mrow + 16 * i, ncol + 16 * j, p.M, p.N, __int_as_float(acc[i][j][0]), __int_as_float(acc[i][j][1]),
                                     __int_as_float(acc[i][j][2]), __int_as_float(acc[i][j][3]));
                else
                    store4<OUT, VEC>(p.C, mrow + 16 * i, ncol + 16 * j, p.M, p.N, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
    } else if constexpr (Cfg::EPI == EPI_ROWCOL) {
        EpiScales<Cfg> sc;
        sc.load(p, mrow, ncol, lsc, mloc, nloc);
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = sc.template value<ORDER>(p, acc[i][j], nullptr, i, j, r);
                store4<OUT, VEC>(p.C, mrow + 16 * i, ncol + 16 * j, p.M, p.N, v[0], v[1], v[2], v[3]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                store4<OUT, VEC>(p.C, mrow + 16 * i, ncol + 16 * j, p.M, p.N, accf[i][j][0], accf[i][j][1], accf[i][j][2], accf[i][j][3]);
    }
}

// 2-byte outputs (bf16 / fp16), N % 8 == 0: stage the wave's WTM x 64 result tile through its own slice of
// LDS (free once the K loop is over) so that global stores are 16 B per lane and whole 128-byte row
// segments per 8 lanes.  A lane's natural 8-byte pieces (4 consecutive n of one m) are store-ISSUE bound:
// 32 B per row and instruction measured 1.6 TB/s; whole-line stores run at the HBM write rate.
// LDS image per wave: [WTM rows][128 B], 16-byte chunk c of row m kept at c ^ (m & 7) (conflict-free
// ds_read_b128, 2-way ds_write_b64).
template <class Cfg, int OUT, int ORDER = 0>
__device__ __forceinline__ void epilogue_lds(const GemmParams &p, typename AccT<Cfg::OP>::type (&acc)[Cfg::MT][Cfg::NTL],
                                             v4f (*accf)[Cfg::EPI == EPI_TILE ? Cfg::NTL : 1], char *smem, int wave, int lane,
                                             int64_t mwave, int64_t nwave, const float *lsc = nullptr, int mlw = 0, int nlw = 0) {
    static_assert(Cfg::WTN == 64, "LDS-staged epilogue assumes a 64-column wave tile (128-byte rows)");
    static_assert(Cfg::NW * Cfg::WTM * 128 <= Cfg::LDS_BYTES, "result tile must fit the operand buffers");
    char *const region = smem + wave * (Cfg::WTM * 128);
    const int g = lane >> 4, ml = lane & 15;
    EpiScales<Cfg> sc;
    sc.load(p, mwave + ml, nwave + 4 * g, lsc, mlw + ml, nlw + 4 * g);
    // store side: lane -> (row it * 8 + (lane >> 3), 16-byte chunk lane & 7); one 64-bit address per lane, advanced by a
    // wave-uniform 8 * N elements per store (per-store m * N + n in 64 bits cost ~10 VALU each)
    const int srow = lane >> 3, sc8 = lane & 7;
    const int64_t rows_left64 = p.M - mwave - srow;
    const int rows_left = rows_left64 > Cfg::WTM ? Cfg::WTM : (int)rows_left64;   // store row it * 8 is valid iff it * 8 < rows_left
    const bool col_ok = nwave + 8 * sc8 < p.N;
    unsigned short *const cbase = (unsigned short *)p.C + (mwave + srow) * p.N + nwave + 8 * sc8;
    const int64_t cstep = 8 * p.N;
    // Row block i (16 rows) is complete after its NTL tiles.  Its two 16-byte read-backs are issued right behind its writes (same wave,
    // same slice: LDS executes them in order), but its two global stores go out one row block LATER, behind the conversion of block i + 1:
    // with one wave per SIMD nothing else hides the LDS round trip of every block (round 3; ~0.5 us per 128-row half before).
    v4i d[2];
#pragma unroll
    for (int i = 0; i < Cfg::MT; ++i) {
#pragma unroll
        for (int j = 0; j < Cfg::NTL; ++j) {
            const v2f v01 = sc.template value2<ORDER>(p, acc[i][j], Cfg::EPI == EPI_TILE ? &accf[i][j] : nullptr, i, j, 0);
            const v2f v23 = sc.template value2<ORDER>(p, acc[i][j], Cfg::EPI == EPI_TILE ? &accf[i][j] : nullptr, i, j, 2);
            v2i o;
            if constexpr (OUT == GNK_BF16) {
                o[0] = (int)gnk_pack2_bf16(v01[0], v01[1]);
                o[1] = (int)gnk_pack2_bf16(v23[0], v23[1]);
            } else {
                o[0] = (int)gnk_pack2_f16(v01[0], v01[1]);
                o[1] = (int)gnk_pack2_f16(v23[0], v23[1]);
            }
            const int row = 16 * i + ml;
            *(GNK_LDS v2i *)(GNK_LDS char *)(region + row * 128 + (((2 * j + (g >> 1)) ^ (ml & 7)) << 4) + 8 * (g & 1)) = o;
        }
        if (i > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = 2 * (i - 1) + u;
                if (col_ok && it * 8 < rows_left) gnk_store16_out(cbase + it * cstep, d[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (2 * i + u) * 8 + srow;
            d[u] = *(const GNK_LDS v4i *)(const GNK_LDS char *)(region + row * 128 + ((sc8 ^ (row & 7)) << 4));
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = 2 * (Cfg::MT - 1) + u;
        if (col_ok && it * 8 < rows_left) gnk_store16_out(cbase + it * cstep, d[u]);
    }
}

// Epilogue entry shared by the kernels.  All LDS tile reads of the workgroup must have retired
// (barrier) before the call.
template <class Cfg>
__device__ __forceinline__ void run_epilogue(const GemmParams &p, typename AccT<Cfg::OP>::type (&acc)[Cfg::MT][Cfg::NTL],
                                             v4f (*accf)[Cfg::EPI == EPI_TILE ? Cfg::NTL : 1], char *smem, int wave, int lane,
                                             int64_t mwave, int64_t nwave, const float *lsc = nullptr, int mlw = 0, int nlw = 0) {
    // lsc: prefetched scales in LDS (ScalePrefetch); mlw / nlw: the wave tile's origin inside the block tile
    const int64_t mrow = mwave + (lane & 15), ncol = nwave + 4 * (lane >> 4);
    const int mloc = mlw + (lane & 15), nloc = nlw + 4 * (lane >> 4);
    const bool aligned16 = ((uintptr_t)p.C & 15) == 0;
    const bool vec_ok = (p.N % 4 == 0) && aligned16;
    if constexpr (Cfg::EPI == EPI_PLAIN && Cfg::OP == OP_I8) {
        if (vec_ok) epilogue<Cfg, GNK_I32, true>(p, acc, accf, mrow, ncol);
        else epilogue<Cfg, GNK_I32, false>(p, acc, accf, mrow, ncol);
    } else {
        const bool lds_ok = (p.N % 8 == 0) && aligned16;
        if (p.out_dtype == GNK_BF16) {
            if (lds_ok) {
                if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue_lds<Cfg, GNK_BF16, 0>(p, acc, accf, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
                else epilogue_lds<Cfg, GNK_BF16, 1>(p, acc, accf, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
            }
            else if (vec_ok) { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<Cfg, GNK_BF16, true, 0>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); else epilogue<Cfg, GNK_BF16, true, 1>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); }
            else { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<Cfg, GNK_BF16, false, 0>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); else epilogue<Cfg, GNK_BF16, false, 1>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); }
        } else if (p.out_dtype == GNK_F16) {
            if (lds_ok) {
                if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue_lds<Cfg, GNK_F16, 0>(p, acc, accf, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
                else epilogue_lds<Cfg, GNK_F16, 1>(p, acc, accf, smem, wave, lane, mwave, nwave, lsc, mlw, nlw);
            }
            else if (vec_ok) { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<Cfg, GNK_F16, true, 0>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); else epilogue<Cfg, GNK_F16, true, 1>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); }
            else { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<Cfg, GNK_F16, false, 0>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); else epilogue<Cfg, GNK_F16, false, 1>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); }
        } else {
            if (vec_ok) { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<Cfg, GNK_F32, true, 0>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); else epilogue<Cfg, GNK_F32, true, 1>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); }
            else { if (Cfg::EPI != EPI_ROWCOL || p.order == 0) epilogue<Cfg, GNK_F32, false, 0>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); else epilogue<Cfg, GNK_F32, false, 1>(p, acc, accf, mrow, ncol, lsc, mloc, nloc); }
        }
    }
}

// ------------------------------------------------------------------------------ tile-scale indexing
// Element offsets of this lane's scale blocks, computed ONCE (divisions by QM / QN per K-step, and dtype
// branches between dependent loads, made the first tile-scaled kernel 10x slower than the row-scaled one).
// The fast path requires QN % 4 == 0 and N % 4 == 0 (a lane's 4 consecutive n share one block) and scale
// offsets < 2^31 (host-checked).  Scales are fetched as RAW bits one K-step ahead and converted at use,
// so their latency hides under a whole step of MFMAs instead of stalling the step that needs them.
template <class Cfg> struct TileScaleIdx {
    static constexpr int MT = Cfg::EPI == EPI_TILE ? Cfg::MT : 1, NTL = Cfg::EPI == EPI_TILE ? Cfg::NTL : 1;
    int aoff[MT];
    int boff[NTL];
    int astep, bstep;
    __device__ __forceinline__ void init(const GemmParams &p, int64_t mrow, int64_t ncol) {
        const unsigned qm = (unsigned)p.QM, qn = (unsigned)p.QN;   // M, N < 2^31 on this path: 32-bit division
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int64_t m = mrow + 16 * i;
            m = m < p.M ? m : p.M - 1;
            aoff[i] = (int)(((unsigned)m / qm) * (unsigned)p.ssam);
        }
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            int64_t n = ncol + 16 * j;
            n = n < p.N ? n : p.N - 1;
            boff[j] = (int)(((unsigned)n / qn) * (unsigned)p.ssbn);
        }
        astep = (int)p.ssak;
        bstep = (int)p.ssbk;
    }
    // one load per scale (these loads share the vector-memory queue with the LDS-DMA stream: doubling them cost 10-40 %);
    // the width is a two-armed uniform branch, the conversion (convert) is branch-free
    __device__ __forceinline__ void load_raw(const GemmParams &p, int t, unsigned (&ra)[MT], unsigned (&rb)[NTL]) const {
        const int at = t * astep, bt = t * bstep;
        if (p.scale_dtype == GNK_F32) {
#pragma unroll
            for (int i = 0; i < MT; ++i) ra[i] = gnk_load_raw<GNK_F32>(p.s0, aoff[i] + at);
#pragma unroll
            for (int j = 0; j < NTL; ++j) rb[j] = gnk_load_raw<GNK_F32>(p.s1, boff[j] + bt);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) ra[i] = gnk_load_raw<GNK_BF16>(p.s0, aoff[i] + at);
#pragma unroll
            for (int j = 0; j < NTL; ++j) rb[j] = gnk_load_raw<GNK_BF16>(p.s1, boff[j] + bt);
        }
    }
    // B side alone (the A side comes from the LDS panel, see gemm_kernel)
    __device__ __forceinline__ void load_raw_b(const GemmParams &p, int t, unsigned (&rb)[NTL]) const {
        const int bt = t * bstep;
        if (p.scale_dtype == GNK_F32) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) rb[j] = gnk_load_raw<GNK_F32>(p.s1, boff[j] + bt);
        } else {
#pragma unroll
            for (int j = 0; j < NTL; ++j) rb[j] = gnk_load_raw<GNK_BF16>(p.s1, boff[j] + bt);
        }
    }
    static __device__ __forceinline__ void convert_b(const GemmParams &p, const unsigned (&rb)[NTL], float (&sbv)[NTL]) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) sbv[j] = gnk_raw_to_f32_rt(rb[j], p.scale_dtype);
    }
    static __device__ __forceinline__ void convert(const GemmParams &p, const unsigned (&ra)[MT], const unsigned (&rb)[NTL], float (&sav)[MT], float (&sbv)[NTL]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) sav[i] = gnk_raw_to_f32_rt(ra[i], p.scale_dtype);
#pragma unroll
        for (int j = 0; j < NTL; ++j) sbv[j] = gnk_raw_to_f32_rt(rb[j], p.scale_dtype);
    }
};

// ------------------------------------------------------------------------------ the kernel
// amdgpu_waves_per_eu(2): the two-buffer kernels are designed for two workgroups per CU (64 KiB of LDS each), i.e. a
// 256-register budget (VGPRs + AGPRs).  Without the hint the int8 tile-scaled instances took 264 and ran at half occupancy.
template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) __attribute__((amdgpu_waves_per_eu(Cfg::NT == 256 ? 2 : 1))) gemm_kernel(const GemmParams p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = Cfg::MT, NTL = Cfg::NTL;
    typedef typename AccT<Cfg::OP>::type acc_t;

    // split-K (plain int32 output only): split s = blockIdx.y covers k in [s * kchunk, min(K, (s+1) * kchunk)) and
    // owns slab s of the workspace; integer partial sums add exactly, so the reduction kernel's result is bit-identical
    // to the unsplit product.  All of this is wave-uniform scalar arithmetic.
    GemmParams p = p_in;
    if constexpr (Cfg::EPI == EPI_PLAIN && Cfg::OP == OP_I8) {
        if (p.kchunk > 0) {
            const int64_t k0 = (int64_t)blockIdx.y * p.kchunk;
            const int64_t kn = p_in.K - k0 < p.kchunk ? p_in.K - k0 : p.kchunk;
            p.K = kn > 0 ? kn : 0;
            p.A += Cfg::A_KMAJOR ? k0 : k0 * p.lda;
            p.B += Cfg::B_KMAJOR ? k0 : k0 * p.ldb;
            p.C = (char *)p.C + (int64_t)blockIdx.y * p.M * p.N * 4;
        }
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    int tm, tn;
    tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;

    // LDS carve: A buffers 0/1, then B buffers 0/1 (one dynamic array: see the guide's note on
    // a second __shared__ object de-pipelining LDS-DMA loops)
    char *const As0 = smem;
    char *const Bs0 = smem + 2 * Cfg::A_BYTES;

    ScalePrefetch<Cfg::BM, Cfg::BN, Cfg::NT, Cfg::EPI> spf;
    spf.issue(p, m0, n0);

    Stager<Cfg::A_KMAJOR, Cfg::BM, Cfg::NW> sa;
    Stager<Cfg::B_KMAJOR, Cfg::BN, Cfg::NW> sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);

    acc_t acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};

    // output coordinates of this lane: m = mrow + 16 i, n = ncol + 16 j + (0..3)
    const int64_t mrow = m0 + wm * Cfg::WTM + (lane & 15);
    const int64_t ncol = n0 + wn * Cfg::WTN + 4 * (lane >> 4);

    // tile-scaled: fp32 running sum + per-K-block scales
    v4f accf[Cfg::EPI == EPI_TILE ? MT : 1][Cfg::EPI == EPI_TILE ? NTL : 1];
    if constexpr (Cfg::EPI == EPI_TILE) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    }

    // Tile scales.  B side: the lane's NTL scale blocks, fetched one K-step ahead (usually one broadcast value: QN >= the tile).
    // A side: per-lane fetches of sA[m / QM, t] touch 16 rows per load instruction (row-major sA: 16 cache lines for 32 bytes)
    // and cost 15-25 % of the kernel; instead the workgroup stages a PANEL of its BM rows x 8 K-blocks in LDS (fp32, [step][row])
    // with one element per thread and load, k fastest (coalesced for row-major sA), once per 8 steps.
    // Scale blocks of Cfg::KB k: NB per 128-k step.  "Scale step" u = t * NB + b indexes sA[:, u] / sB[u, :]; the panel holds 8 of them.
    constexpr int TS_STEPS = 8, TS_PE = Cfg::EPI == EPI_TILE ? Cfg::BM * TS_STEPS / Cfg::NT : 1;
    constexpr bool MULTI = Cfg::KB > GNK_BK;                    // scale blocks of kgroup * 128 k (its own instances: the run-time group
                                                                // logic cost the plain 128-k kernel 4 %)
    constexpr int NB = MULTI ? 1 : GNK_BK / Cfg::KB, PS = TS_STEPS / NB;    // PS: K-steps per panel
    // NB > 1: the B-side scales go through a second panel ([scale step][BN columns]) too -- NB x NTL per-lane values for this step
    // and as many in flight for the next one spilled registers.
    constexpr bool PANEL_B = NB > 1 || Cfg::PERCOL;
    constexpr int TS_PEB = PANEL_B ? Cfg::BN * TS_STEPS / Cfg::NT : 1;
    const int nk = (int)((p.K + GNK_BK - 1) / GNK_BK);          // the last step may be partial (plain / row-col epilogues; host-checked)
    const int ktail = (int)(p.K % GNK_BK);                        // k elements of that partial step (0: none)
    const int G = MULTI && p.kgroup > 1 ? p.kgroup : 1;                             // K-steps per scale block (QK = G * 128)
    const int nks = (nk * NB + G - 1) / G;                                          // scale steps along K
    const int psteps = PS * G;                                                      // K-steps covered by one panel
    TileScaleIdx<Cfg> ts;
    unsigned raw_b[Cfg::EPI == EPI_TILE && !PANEL_B ? NTL : 1], praw[TS_PE], prawb[TS_PEB];
    int poff[TS_PE], poffb[TS_PEB];
    float *const panel = (float *)(smem + Cfg::LDS_BYTES);
    float *const panel_b = panel + Cfg::BM * TS_STEPS;
    auto panel_fetch = [&](int pn) {
#pragma unroll
        for (int u = 0; u < TS_PE; ++u) {
            const int tt = pn * TS_STEPS + (((int)threadIdx.x + Cfg::NT * u) % TS_STEPS);
            praw[u] = tt < nks ? gnk_load_raw_rt(p.s0, p.scale_dtype, poff[u] + tt * (int)p.ssak) : 0u;
        }
        if constexpr (PANEL_B) {
#pragma unroll
            for (int u = 0; u < TS_PEB; ++u) {
                const int tt = pn * TS_STEPS + (((int)threadIdx.x + Cfg::NT * u) % TS_STEPS);
                prawb[u] = tt < nks ? gnk_load_raw_rt(p.s1, p.scale_dtype, poffb[u] + tt * (int)p.ssbk) : 0u;
            }
        }
    };
    auto panel_store = [&]() {
#pragma unroll
        for (int u = 0; u < TS_PE; ++u) {
            const int e = (int)threadIdx.x + Cfg::NT * u;
            *(GNK_LDS float *)(GNK_LDS char *)((char *)panel + ((e % TS_STEPS) * Cfg::BM + e / TS_STEPS) * 4) = gnk_raw_to_f32_rt(praw[u], p.scale_dtype);
        }
        if constexpr (PANEL_B) {
#pragma unroll
            for (int u = 0; u < TS_PEB; ++u) {
                const int e = (int)threadIdx.x + Cfg::NT * u;
                *(GNK_LDS float *)(GNK_LDS char *)((char *)panel_b + ((e % TS_STEPS) * Cfg::BN + e / TS_STEPS) * 4) = gnk_raw_to_f32_rt(prawb[u], p.scale_dtype);
            }
        }
    };
    if constexpr (Cfg::EPI == EPI_TILE) {
        ts.init(p, mrow, ncol);
#pragma unroll
        for (int u = 0; u < TS_PE; ++u) {
            int64_t m = m0 + ((int)threadIdx.x + Cfg::NT * u) / TS_STEPS;
            m = m < p.M ? m : p.M - 1;
            poff[u] = (int)(((unsigned)m / (unsigned)p.QM) * (unsigned)p.ssam);
        }
        if constexpr (PANEL_B) {
#pragma unroll
            for (int u = 0; u < TS_PEB; ++u) {
                int64_t n = n0 + ((int)threadIdx.x + Cfg::NT * u) / TS_STEPS;
                n = n < p.N ? n : p.N - 1;
                poffb[u] = (int)(((unsigned)n / (unsigned)p.QN) * (unsigned)p.ssbn);
            }
        }
        if (nk > 0) {
            if constexpr (!PANEL_B) ts.load_raw_b(p, 0, raw_b);
            panel_fetch(0);
        }
    }

    if (nk > 0) {
        if (nk == 1 && ktail) {
            sa.issue_tail(As0, p.lda, m0, p.M, wave, lane, ktail);
            sb.issue_tail(Bs0, p.ldb, n0, p.N, wave, lane, ktail);
        } else {
            sa.issue(As0, wave);
            sb.issue(Bs0, wave);
        }
    }
    // scale bookkeeping as running counters (wave-uniform; divisions by the run-time G cost ~5 % of the tile-scaled fp8 kernel):
    // c_gph = t % G, c_u = t / G (scale step), c_tk = t % psteps, c_tsi = c_tk / G, c_pn = t / psteps (panel number)
    int c_gph = 0, c_u = 0, c_tk = 0, c_tsi = 0, c_pn = 0;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // tile-scaled: this K-block's scales arrived with the vmcnt(0) above; fetch the next block's BEFORE
        // the next stage is issued (they are waited for together at the top of the next step)
        float sbv1[Cfg::EPI == EPI_TILE && !PANEL_B ? NTL : 1];
        int tsi = 0;
        if constexpr (Cfg::EPI == EPI_TILE) {
            tsi = c_tsi;               // scale step of this K-step within the panel (times NB)
            if (c_tk == 0) {           // the panel of the next 8 scale steps was fetched during the step before it (or before the loop)
                panel_store();
                __syncthreads();
            }
            if constexpr (!PANEL_B) {
                TileScaleIdx<Cfg>::convert_b(p, raw_b, sbv1);
                if (t + 1 < nk) ts.load_raw_b(p, c_u + (c_gph + 1 == G ? 1 : 0), raw_b);
            }
            if (c_tk == psteps - 1 && t + 1 < nk) panel_fetch(c_pn + 1);
        }

        const int cur = t & 1;
        if constexpr (Cfg::EPI != EPI_TILE) {
            if (ktail && t + 1 == nk) {      // partial last step: clear k >= ktail of both tiles (wave-uniform branch, once per kernel)
                zero_tail<Cfg::A_KMAJOR, Cfg::BM, Cfg::NT>(As0 + cur * Cfg::A_BYTES, ktail);
                zero_tail<Cfg::B_KMAJOR, Cfg::BN, Cfg::NT>(Bs0 + cur * Cfg::B_BYTES, ktail);
                __syncthreads();
            }
        }
        if (t + 1 < nk) {
            if (Cfg::EPI != EPI_TILE && ktail && t + 2 == nk) {
                sa.issue_tail(As0 + (cur ^ 1) * Cfg::A_BYTES, p.lda, m0, p.M, wave, lane, ktail);
                sb.issue_tail(Bs0 + (cur ^ 1) * Cfg::B_BYTES, p.ldb, n0, p.N, wave, lane, ktail);
            } else {
                sa.issue(As0 + (cur ^ 1) * Cfg::A_BYTES, wave);
                sb.issue(Bs0 + (cur ^ 1) * Cfg::B_BYTES, wave);
            }
        }
        if constexpr (Cfg::EPI != EPI_TILE) {
            compute_step<Cfg, false>(As0 + cur * Cfg::A_BYTES, Bs0 + cur * Cfg::B_BYTES, acc, wm, wn, lane);
        } else {
            // NB > 1: a REAL loop over the step's blocks.  Every block starts from C = 0, so nothing orders block b + 1's MFMAs after
            // block b's rescale; unrolled, the scheduler issued all four blocks' MFMAs first (four accumulator sets live: 160 spills).
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                if constexpr (NB == 1) {
                    // QK = G * 128: the partial sum runs over G steps (exact in int32) and is rescaled after the last of them
                    if (!MULTI || c_gph == 0) compute_step<Cfg, true>(As0 + cur * Cfg::A_BYTES, Bs0 + cur * Cfg::B_BYTES, acc, wm, wn, lane);
                    else compute_step<Cfg, false>(As0 + cur * Cfg::A_BYTES, Bs0 + cur * Cfg::B_BYTES, acc, wm, wn, lane);
                    if (MULTI && c_gph + 1 != G) continue;
                } else if constexpr (Cfg::OP == OP_I8 && Cfg::KB == 64) {
                    // int8, 64-k blocks: one 64-k MFMA per tile on the usual 16-byte fragments of half b (the 32-k MFMA occupies the
                    // matrix pipe just as long for half the work)
                    const char *As = As0 + cur * Cfg::A_BYTES, *Bs = Bs0 + cur * Cfg::B_BYTES;
                    v4i fa[MT], fb[NTL];
#pragma unroll
                    for (int j = 0; j < NTL; ++j) fb[j] = load_frag<Cfg::B_KMAJOR, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, b, lane);
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa[i] = load_frag<Cfg::A_KMAJOR, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, b, lane);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[j], fa[i], (v4i){0, 0, 0, 0}, 0, 0, 0);
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < Cfg::KB / 32; ++s2)
                        compute_sub32<Cfg>(As0 + cur * Cfg::A_BYTES, Bs0 + cur * Cfg::B_BYTES, acc, wm, wn, lane, b * (Cfg::KB / 32) + s2, s2 == 0);
                }
                // this block's scales from the panel(s) (fp32, [scale step][row] / [scale step][column])
                float sav[MT], sbv[NTL];
                v4f sbq[Cfg::PERCOL ? NTL : 1];     // PERCOL: one scale per column of the lane's quad
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    sav[i] = *(const GNK_LDS float *)(const GNK_LDS char *)((const char *)panel + ((tsi * NB + b) * Cfg::BM + wm * Cfg::WTM + 16 * i + (lane & 15)) * 4);
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    if constexpr (Cfg::PERCOL) {
                        sbq[j] = *(const GNK_LDS v4f *)(const GNK_LDS char *)((const char *)panel_b + ((tsi * NB + b) * Cfg::BN + wn * Cfg::WTN + 16 * j + 4 * (lane >> 4)) * 4);
                        sbv[j] = sbq[j][0];
                    } else if constexpr (PANEL_B)
                        sbv[j] = *(const GNK_LDS float *)(const GNK_LDS char *)((const char *)panel_b + ((tsi * NB + b) * Cfg::BN + wn * Cfg::WTN + 16 * j + 4 * (lane >> 4)) * 4);
                    else sbv[j] = sbv1[j];
                }
                // acc_f32 += (f32(partial) * sA[m/QM, u]) * sB[u, n/QN]   (triton_mm.py:327-331)
                if constexpr (Cfg::FUSED) {
                    // tolerance mode (16-bit outputs by default, gnk_set_tile_rescale): one multiply per tile and K-block for s = sA * sB, then
                    // convert + v_fma_f32 per element -- 2 VALU operations per element and block instead of 4 (the VALU stream, not the
                    // MFMAs, bounds this kernel: DESIGN 6.1a).  An NVIDIA build of the reference contracts its last multiply-add too
                    // (SURVEY 9.5); parity is the reference's tolerance (benchmark_mm.py:119-121, 144-145), not the interpreter's bits.
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            const float s = __fmul_rn(sav[i], sbv[j]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float part;
                                if constexpr (Cfg::OP == OP_I8) part = (float)acc[i][j][r];
                                else part = acc[i][j][r];
                                if constexpr (Cfg::PERCOL) accf[i][j][r] = __builtin_fmaf(part, __fmul_rn(sav[i], sbq[j][r]), accf[i][j][r]);
                                else accf[i][j][r] = __builtin_fmaf(part, s, accf[i][j][r]);
                            }
                        }
                } else {
                    // the Triton interpreter's form, bit for bit: three separately rounded operations, no FMA
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float part;
                                if constexpr (Cfg::OP == OP_I8) part = (float)acc[i][j][r];
                                else part = acc[i][j][r];
                                float tmp = __fmul_rn(part, sav[i]);
                                if constexpr (Cfg::PERCOL) tmp = __fmul_rn(tmp, sbq[j][r]);
                                else tmp = __fmul_rn(tmp, sbv[j]);
                                accf[i][j][r] = __fadd_rn(accf[i][j][r], tmp);
                            }
                        }
                }
            }
            if (++c_gph == G) { c_gph = 0; ++c_u; ++c_tsi; }
            if (++c_tk == psteps) { c_tk = 0; c_tsi = 0; ++c_pn; }
        }
    }

    // ------------------------------------------------------------------ epilogue
    spf.park(p, smem + Cfg::LDS_BYTES);
    __syncthreads();   // every wave is done reading the operand tiles: LDS is free for the result staging; scales are parked
    run_epilogue<Cfg>(p, acc, accf, smem, wave, lane, m0 + wm * Cfg::WTM, n0 + wn * Cfg::WTN,
                      Cfg::EPI == EPI_ROWCOL ? (const float *)(smem + Cfg::LDS_BYTES) : nullptr, wm * Cfg::WTM, wn * Cfg::WTN);
}

template <class Cfg> int launch(const GemmParams &p0, hipStream_t st) {
    constexpr int LDS_TOTAL = Cfg::LDS_BYTES + ScalePrefetch<Cfg::BM, Cfg::BN, Cfg::NT, Cfg::EPI>::BYTES + (Cfg::EPI == EPI_TILE ? Cfg::BM * 8 * 4 : 0) +
                              (Cfg::EPI == EPI_TILE && (Cfg::KB < GNK_BK || Cfg::PERCOL) ? Cfg::BN * 8 * 4 : 0);   // tile-scaled: A-side scale panel (+ B-side panel for 64 / 32-k blocks and per-column scales)
    GemmParams p = p0;
    p.grid_m = (int)((p.M + Cfg::BM - 1) / Cfg::BM);
    p.grid_n = (int)((p.N + Cfg::BN - 1) / Cfg::BN);
    static std::atomic<bool> attr_set[64];   // opt in to > 64 KiB of dynamic LDS once per device and kernel instance
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "gemm: no current HIP device");
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void *)gemm_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return gnk_fail(GNK_ERR_LAUNCH, "gemm: cannot reserve %d bytes of LDS", LDS_TOTAL);
        attr_set[dev].store(true, std::memory_order_release);
    }
    const unsigned splits = p.kchunk > 0 ? (unsigned)((p.K + p.kchunk - 1) / p.kchunk) : 1u;
    hipLaunchKernelGGL((gemm_kernel<Cfg>), dim3((unsigned)(p.grid_m * p.grid_n), splits), dim3(Cfg::NT), LDS_TOTAL, st, p);
    return gnk_check_launch("gemm");
}

}  // namespace gnk_gemm
