// gnk_quantize.hip -- absmax symmetric INT8 quantization for gfx950 (HBM-bound).
//
// Restates subclasses/int8.py:12-26 of the reference as hand-written kernels:
//   * rowwise  : reduction along the contiguous dim. One wave per row; the row is read ONCE with
//                16-byte loads and kept in registers between the absmax and the quantize step
//                (algorithmic traffic 3 B/elem for bf16 -> int8).
//   * colwise  : reduction along the strided dim (the grad-input / grad-weight operands of
//                subclasses/mixed_precision.py:178,186).  Pass 1: per-column absmax into an fp32
//                workspace with one atomicMax per block and column; pass 2: quantize in place of
//                the reference's strided reduction + transpose copy. Output keeps x's layout, so
//                the GEMM consumes it through its transposed-operand (ds_read_b64_tr_b8) path.
//   * generic  : any strides / alignment / dtype, one block per logical row (slow, correct).
#include "gnk_common.h"

#include <mutex>

namespace {

// tuning knobs (A/B builds: tools/quant_ab.sh)
#ifndef GNK_RC_PREFETCH_BELOW
#define GNK_RC_PREFETCH_BELOW 12     // quant_rowwise_colmax_kernel prefetches the next row while NCH < this
#endif
#ifndef GNK_RC_CAP
#define GNK_RC_CAP 2                 // its persistent grid: blocks (8 waves) per CU
#endif
#ifndef GNK_COLQ_WIDE
#define GNK_COLQ_WIDE 1               // column-wise pass with a given absmax: the one-row-lane shape for wide matrices
#endif
#ifndef GNK_COLQ_PB
#define GNK_COLQ_PB 128              // rows per block of the column-wise pass with a given absmax (large matrices)
#endif

template <int DT> struct ElemTraits;
template <> struct ElemTraits<GNK_BF16> { static constexpr int VEC = 8; };
template <> struct ElemTraits<GNK_F16>  { static constexpr int VEC = 8; };
template <> struct ElemTraits<GNK_F32>  { static constexpr int VEC = 4; };

// unpack one 16-byte chunk into VEC floats
template <int DT> __device__ __forceinline__ void unpack16(const v4i &raw, float *out) {
    if constexpr (DT == GNK_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = __int_as_float(raw[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned w = (unsigned)raw[i];
            if constexpr (DT == GNK_BF16) {
                out[2 * i] = __uint_as_float(w << 16);
                out[2 * i + 1] = __uint_as_float(w & 0xffff0000u);
            } else {
                out[2 * i] = gnk_f16_to_f32((unsigned short)(w & 0xffffu));
                out[2 * i + 1] = gnk_f16_to_f32((unsigned short)(w >> 16));
            }
        }
    }
}

// wave-wide maximum of non-negative fp32 BIT PATTERNS (integer max: NaN and Inf keep their order, see GnkAbsMax)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned pack4_i8(int a, int b, int c, int d) {
    return (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)(d & 0xff) << 24);
}

// ------------------------------------------------------------------------------------------
// rowwise: one wave per row, NCH 16-byte chunks per lane held in registers.
// Requirements (checked on the host): sxc == 1, sqc == 1, C % VEC == 0, row starts 16-B aligned
// for x and VEC-byte aligned for q, C <= 64 * VEC * NCH.
// ------------------------------------------------------------------------------------------
// FAST: the row's divisor passed GnkDivisor::ok() (wave-uniform), so the shared-reciprocal division applies.
template <int DT, bool SR, int VEC, bool NT = false>
__device__ __forceinline__ void quant_chunk(const v4i &raw, const GnkDivisor &dv, bool fast, int8_t *qdst, unsigned long long seed,
                                            unsigned long long offset, unsigned long long idx0) {
    float f[VEC];
    unpack16<DT>(raw, f);
    int qi[VEC];
    if constexpr (SR) {
        if (((idx0 + offset) & 3ull) == 0) {      // the usual case (rows of a multiple of 4 elements): one Philox block per 4 elements
#pragma unroll
            for (int g = 0; g < VEC / 4; ++g) {
                float u[4];
                gnk_sr_uniform4(seed, offset, idx0 + 4 * g, u);
#pragma unroll
                for (int k = 0; k < 4; ++k) qi[4 * g + k] = gnk_sr_clip_i8(fast ? dv.quot(f[4 * g + k]) : f[4 * g + k] / dv.d, u[k]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) qi[j] = gnk_quant_one<true>(f[j], dv.d, seed, offset, idx0 + j);
        }
    } else {
        if (fast) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) qi[j] = gnk_rne_clip_i8(dv.quot(f[j]));
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) qi[j] = gnk_rne_clip_i8_any(f[j] / dv.d);
        }
    }
    if constexpr (VEC == 8) {
        v2i o = {(int)pack4_i8(qi[0], qi[1], qi[2], qi[3]), (int)pack4_i8(qi[4], qi[5], qi[6], qi[7])};
        if constexpr (NT) __builtin_nontemporal_store(o, (v2i *)qdst);
        else *(v2i *)qdst = o;
    } else {
        *(unsigned *)qdst = pack4_i8(qi[0], qi[1], qi[2], qi[3]);
    }
}

// Each wave walks rows wave_id, wave_id + nwaves, ... and loads row i+1 while it reduces, quantizes and stores
// row i: with one row per wave every wave of the chip is in the same phase at the same time (all loading, then
// all computing, then all storing) and HBM idles two thirds of the time.
template <int DT, int NCH, bool SR, bool NT>
__global__ void __launch_bounds__(256)
quant_rowwise_kernel(const char *__restrict__ x, int8_t *__restrict__ q, void *__restrict__ scale,
                     int64_t R, int64_t C, int64_t sxr_bytes, int64_t sqr, float eps,
                     unsigned long long seed, unsigned long long offset) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;                                      // wave-uniform
    const int nchunks = (int)(C / VEC);

    v4i raw[NCH], nxt[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = i * 64 + lane;
        nxt[i] = (v4i){0, 0, 0, 0};
        if (ch < nchunks) nxt[i] = *(const v4i *)(x + r * sxr_bytes + (int64_t)ch * 16);
    }
    for (; r < R; r += nwaves) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) raw[i] = nxt[i];
        const int64_t rn = r + nwaves;
        if (rn < R) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = i * 64 + lane;
                if (ch < nchunks) nxt[i] = *(const v4i *)(x + rn * sxr_bytes + (int64_t)ch * 16);
            }
        }
        GnkAbsMax<DT> am;
#pragma unroll
        for (int i = 0; i < NCH; ++i) am.add16(raw[i]);
        const float amax = __uint_as_float(wave_umax(am.bits()));
        const float s = amax / 127.0f;
        GnkDivisor dv;
        dv.set(gnk_clamp_min_nan(s, eps));
        const bool fast = dv.ok();
        if (lane == 0) gnk_store_scalar_rt(scale, DT, r, s);

        int8_t *qrow = q + r * sqr;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = i * 64 + lane;
            if (ch < nchunks)
                quant_chunk<DT, SR, VEC, NT>(raw[i], dv, fast, qrow + (int64_t)ch * VEC, seed, offset,
                                             (unsigned long long)(r * C + (int64_t)ch * VEC));
        }
    }
}

// rowwise for very long rows: one block (256 threads) per row, two passes (second pass re-reads
// the row, which is L2-resident).
template <int DT, bool SR>
__global__ void __launch_bounds__(256)
quant_rowwise_long_kernel(const char *__restrict__ x, int8_t *__restrict__ q, void *__restrict__ scale,
                          int64_t R, int64_t C, int64_t sxr_bytes, int64_t sqr, float eps,
                          unsigned long long seed, unsigned long long offset) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    __shared__ unsigned red[4];
    const int64_t r = blockIdx.x;
    const int64_t nchunks = C / VEC;
    const char *xrow = x + r * sxr_bytes;
    GnkAbsMax<DT> am;
    for (int64_t ch = threadIdx.x; ch < nchunks; ch += 256) am.add16(*(const v4i *)(xrow + ch * 16));
    const unsigned wmax = wave_umax(am.bits());
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
    __syncthreads();
    const unsigned m01 = red[0] > red[1] ? red[0] : red[1], m23 = red[2] > red[3] ? red[2] : red[3];
    const float amax = __uint_as_float(m01 > m23 ? m01 : m23);
    const float s = amax / 127.0f;
    GnkDivisor dv;
    dv.set(gnk_clamp_min_nan(s, eps));
    const bool fast = dv.ok();
    if (threadIdx.x == 0) gnk_store_scalar_rt(scale, DT, r, s);
    int8_t *qrow = q + r * sqr;
    for (int64_t ch = threadIdx.x; ch < nchunks; ch += 256)
        quant_chunk<DT, SR, VEC>(*(const v4i *)(xrow + ch * 16), dv, fast, qrow + ch * VEC, seed, offset,
                                 (unsigned long long)(r * C + ch * VEC));
}

// ------------------------------------------------------------------------------------------
// fused: row-wise quantize + per-column absmax of the SAME read.
// subclasses/mixed_precision.py quantizes every X, W and dY twice per step, once along each dim
// (:100-108 with :154,:178,:186).  The column-wise pass needs the column absmax before it can
// quantize anything, i.e. two reads; this kernel collects it while the row-wise pass has the row
// in registers anyway, so the column-wise quantization later costs ONE more read instead of two.
//   * a lane owns the same columns in every row it sees (chunk i * 64 + lane), so the running
//     column maxima are per-lane registers: packed |x| bit patterns for 16-bit inputs
//     (v_pk_max_u16 on raw & 0x7fff7fff: non-negative floats order like their bits), floats for f32;
//   * no global atomics (every block hitting the same C words serialises in L2: measured 5-20x
//     slower): the block's 8 waves merge through LDS (ds_max_u32 per column) and write ONE partial
//     row; a second, tiny kernel reduces the <= 2 * CUs partial rows to fp32 column maxima.
// ------------------------------------------------------------------------------------------

template <int DT, int NCH, bool SR, bool NT>
__global__ void __launch_bounds__(512)
quant_rowwise_colmax_kernel(const char *__restrict__ x, int8_t *__restrict__ q, void *__restrict__ scale,
                            unsigned *__restrict__ partial, int64_t R, int64_t C, int64_t sxr_bytes, int64_t sqr, float eps,
                            unsigned long long seed, unsigned long long offset) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    extern __shared__ unsigned colmax_lds[];                 // C words: this block's column maxima as fp32 bit patterns
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 8;
    int64_t r = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    const int nchunks = (int)(C / VEC);
    for (int c = threadIdx.x; c < (int)C; c += 512) colmax_lds[c] = 0u;

    unsigned cm[NCH][4];                                     // 16-bit inputs: 2 columns per word; f32: fp32 bits, 1 column per word
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cm[i][j] = 0u;

    // The next row is prefetched while this one is processed, except for the widest rows: three row-sized register sets (row,
    // next row, column maxima: 144 registers at NCH = 12) spilled at the 128 registers that four waves per SIMD leave; there
    // the 16 waves of a CU with 12 loads in flight each hide the latency without a prefetch.
    constexpr bool PREFETCH = NCH < GNK_RC_PREFETCH_BELOW;
    v4i raw[NCH], nxt[PREFETCH ? NCH : 1];
    if constexpr (PREFETCH) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = i * 64 + lane;
            nxt[i] = (v4i){0, 0, 0, 0};
            if (r < R && ch < nchunks) nxt[i] = *(const v4i *)(x + r * sxr_bytes + (int64_t)ch * 16);
        }
    }
    for (; r < R; r += nwaves) {
        if constexpr (PREFETCH) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) raw[i] = nxt[i];
            const int64_t rn = r + nwaves;
            if (rn < R) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int ch = i * 64 + lane;
                    if (ch < nchunks) nxt[i] = *(const v4i *)(x + rn * sxr_bytes + (int64_t)ch * 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = i * 64 + lane;
                raw[i] = (v4i){0, 0, 0, 0};
                if (ch < nchunks) raw[i] = *(const v4i *)(x + r * sxr_bytes + (int64_t)ch * 16);
            }
        }
        GnkAbsMax<DT> am;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned a = (unsigned)raw[i][j] & (DT == GNK_F32 ? 0x7fffffffu : 0x7fff7fffu);   // |x| bit patterns (1 or 2 per word)
                if constexpr (DT == GNK_F32) { cm[i][j] = a > cm[i][j] ? a : cm[i][j]; am.m = a > am.m ? a : am.m; }
                else { cm[i][j] = gnk_pk_max_u16(cm[i][j], a); am.m = gnk_pk_max_u16(am.m, a); }
            }
        }
        const float amax = __uint_as_float(wave_umax(am.bits()));
        const float s = amax / 127.0f;
        GnkDivisor dv;
        dv.set(gnk_clamp_min_nan(s, eps));
        const bool fast = dv.ok();
        if (lane == 0) gnk_store_scalar_rt(scale, DT, r, s);

        int8_t *qrow = q + r * sqr;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = i * 64 + lane;
            if (ch < nchunks)
                quant_chunk<DT, SR, VEC, NT>(raw[i], dv, fast, qrow + (int64_t)ch * VEC, seed, offset,
                                             (unsigned long long)(r * C + (int64_t)ch * VEC));
        }
    }

    // merge the 8 waves' column maxima (non-negative fp32 bit patterns order like unsigned integers)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = i * 64 + lane;
        if (ch < nchunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (DT == GNK_F32) {
                    if (cm[i][j]) atomicMax(&colmax_lds[ch * 4 + j], cm[i][j]);
                } else {
                    const unsigned lo = cm[i][j] & 0xffffu, hi = cm[i][j] >> 16;
                    unsigned flo, fhi;
                    if constexpr (DT == GNK_BF16) { flo = lo << 16; fhi = hi << 16; }
                    else { flo = __float_as_uint(gnk_f16_to_f32((unsigned short)lo)); fhi = __float_as_uint(gnk_f16_to_f32((unsigned short)hi)); }
                    if (flo) atomicMax(&colmax_lds[ch * 8 + 2 * j], flo);
                    if (fhi) atomicMax(&colmax_lds[ch * 8 + 2 * j + 1], fhi);
                }
            }
        }
    }
    __syncthreads();
    unsigned *prow = partial + (int64_t)blockIdx.x * C;
    for (int c = threadIdx.x * 4; c < (int)C; c += 512 * 4)   // C % 4 == 0 (C % VEC == 0)
        *(v4i *)(prow + c) = *(const v4i *)(colmax_lds + c);
}

// partial[G][C] (fp32 bit patterns) -> colmax[C].  64 columns x 16 row lanes per block, 8 independent loads in flight per
// thread: the kernel is pure latency (a serial walk over 512 partial rows took 25 us; this takes ~3).
__global__ void __launch_bounds__(1024) colmax_reduce_kernel(const unsigned *__restrict__ partial, unsigned *__restrict__ colmax, int64_t G, int64_t C) {
    __shared__ unsigned red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    unsigned m = 0u;
    if (c < C) {
        for (int64_t g0 = ty; g0 < G; g0 += 16 * 8) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t g = g0 + 16 * u;
                v[u] = g < G ? partial[g * C + c] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) m = v[u] > m ? v[u] : m;
        }
    }
    red[ty][tx] = m;
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
        for (int i = 1; i < 16; ++i) m = red[i][tx] > m ? red[i][tx] : m;
        colmax[c] = m;
    }
}

// ------------------------------------------------------------------------------------------
// colwise: memory matrix [P][Q] (Q contiguous); logical rows are the Q columns, the reduction
// runs over P.  Block = 32 column-lanes (VEC columns each) x 8 row-lanes over a PB-row slab.
// ------------------------------------------------------------------------------------------
// PB = rows per block: 128 for large matrices, 32 when 128-row slabs would leave the chip under-filled

template <int DT, int COL_PB>
__global__ void __launch_bounds__(256)
col_absmax_kernel(const char *__restrict__ x, unsigned *__restrict__ amax_ws, int64_t P, int64_t Q,
                  int64_t ld_bytes) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    __shared__ unsigned red[8][32 * VEC + 1];             // fp32 bit patterns of |x| maxima (integer order: NaN > Inf > finite, GnkAbsMax)
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t col0 = ((int64_t)blockIdx.x * 32 + cl) * VEC;
    const int64_t row0 = (int64_t)blockIdx.y * COL_PB;
    unsigned mw[4] = {0u, 0u, 0u, 0u};                     // per input dword: 2 packed 16-bit patterns, or 1 fp32 pattern
    if (col0 < Q) {
        constexpr int NIT = COL_PB / 8;
        v4i raw[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int64_t row = row0 + i * 8 + rl;
            raw[i] = (v4i){0, 0, 0, 0};
            if (row < P) raw[i] = *(const v4i *)(x + row * ld_bytes + col0 * (16 / VEC));
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (DT == GNK_F32) { const unsigned a = (unsigned)raw[i][j] & 0x7fffffffu; mw[j] = a > mw[j] ? a : mw[j]; }
                else mw[j] = gnk_pk_max_u16(mw[j], (unsigned)raw[i][j] & 0x7fff7fffu);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (DT == GNK_F32) red[rl][cl * VEC + j] = mw[j];
        else if constexpr (DT == GNK_BF16) { red[rl][cl * VEC + 2 * j] = (mw[j] & 0xffffu) << 16; red[rl][cl * VEC + 2 * j + 1] = mw[j] & 0xffff0000u; }
        else {
            red[rl][cl * VEC + 2 * j] = __float_as_uint(gnk_f16_to_f32((unsigned short)(mw[j] & 0xffffu)));
            red[rl][cl * VEC + 2 * j + 1] = __float_as_uint(gnk_f16_to_f32((unsigned short)(mw[j] >> 16)));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
        unsigned v = red[0][c];
#pragma unroll
        for (int i = 1; i < 8; ++i) v = red[i][c] > v ? red[i][c] : v;
        const int64_t col = (int64_t)blockIdx.x * 32 * VEC + c;
        if (col < Q && v) atomicMax(amax_ws + col, v);
    }
}

// Pass 1 without atomics and without a cleared workspace (round 2; the hipMemsetAsync in front of every strided reduction was
// ~480 fill launches per Llama-470m step): block (bx, by) owns CL column-lanes x `rpb` rows and writes ONE row of partial maxima,
// partial[by][col]; pass 2 reduces the <= GNK_COL_PARTS partial rows of its columns itself.
// Two geometries: CL = 32 (512-byte row segments, 8 row-lanes, 16 loads of 16 B in flight per thread -- col_absmax_kernel's) when that
// fills the chip, CL = 8 (one 128-byte segment per row, 32 row-lanes, 8 loads in flight) for narrow matrices.
constexpr int GNK_COL_PARTS = 32;
template <int DT, int CL>
__global__ void __launch_bounds__(256)
col_absmax_partial_kernel(const char *__restrict__ x, unsigned *__restrict__ partial, int64_t P, int64_t Q, int64_t ld_bytes, int64_t rpb) {
    constexpr int VEC = ElemTraits<DT>::VEC, RL = 256 / CL, U = CL == 32 ? 16 : 8;
    __shared__ unsigned red[RL][CL * VEC + 1];            // fp32 bit patterns of |x| maxima (integer order: NaN > Inf > finite, GnkAbsMax)
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int64_t col0 = ((int64_t)blockIdx.x * CL + cl) * VEC;
    const int64_t row0 = (int64_t)blockIdx.y * rpb;
    const int64_t rend = row0 + rpb < P ? row0 + rpb : P;
    unsigned mw[4] = {0u, 0u, 0u, 0u};                     // per input dword: 2 packed 16-bit patterns, or 1 fp32 pattern
    if (col0 < Q) {
        const char *src = x + col0 * (16 / VEC);
        for (int64_t r = row0 + rl; r < rend; r += RL * U) {
            v4i raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t row = r + RL * u;
                raw[u] = (v4i){0, 0, 0, 0};
                if (row < rend) raw[u] = *(const v4i *)(src + row * ld_bytes);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (DT == GNK_F32) { const unsigned a = (unsigned)raw[u][j] & 0x7fffffffu; mw[j] = a > mw[j] ? a : mw[j]; }
                    else mw[j] = gnk_pk_max_u16(mw[j], (unsigned)raw[u][j] & 0x7fff7fffu);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (DT == GNK_F32) red[rl][cl * VEC + j] = mw[j];
        else if constexpr (DT == GNK_BF16) { red[rl][cl * VEC + 2 * j] = (mw[j] & 0xffffu) << 16; red[rl][cl * VEC + 2 * j + 1] = mw[j] & 0xffff0000u; }
        else {
            red[rl][cl * VEC + 2 * j] = __float_as_uint(gnk_f16_to_f32((unsigned short)(mw[j] & 0xffffu)));
            red[rl][cl * VEC + 2 * j + 1] = __float_as_uint(gnk_f16_to_f32((unsigned short)(mw[j] >> 16)));
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < CL * VEC; c += 256) {
        unsigned v = red[0][c];
#pragma unroll
        for (int i = 1; i < RL; ++i) v = red[i][c] > v ? red[i][c] : v;
        const int64_t col = (int64_t)blockIdx.x * CL * VEC + c;
        if (col < Q) partial[(int64_t)blockIdx.y * Q + col] = v;
    }
}

// Rows per pass-1 block and the number of partial rows (<= GNK_COL_PARTS) of a reduction over P memory rows, per geometry:
// wide (CL = 32): whole 128-row slabs; narrow (CL = 8): a power of two >= 32.
inline void col_partial_plan(int64_t P, bool wide, int64_t &rpb, int64_t &nparts) {
    if (wide) {
        const int64_t nslabs = (P + 127) / 128;
        rpb = 128 * ((nslabs + GNK_COL_PARTS - 1) / GNK_COL_PARTS);
    } else {
        rpb = 32;
        while ((P + rpb - 1) / rpb > GNK_COL_PARTS) rpb *= 2;
    }
    nparts = (P + rpb - 1) / rpb;
}

// PARTS: amax_ws holds `nparts` partial rows [nparts][Q] (col_absmax_partial_kernel) instead of one reduced row.
// CLN: column-lanes per block (VEC columns each); 256 / CLN row-lanes.  32: 512-byte row segments, 8 row-lanes (round 1).  256: one
// row-lane -- a block walks whole 4-KiB row spans (16-bit inputs) like the row-wise kernel does; for wide matrices the 512-byte segments
// of 128 different rows per block ran at 4.0 TB/s where the row-wise pass reaches 5.6 (round 3, tools/quant_step_probe.py).
template <int DT, bool SR, int COL_PB, bool NT, bool PARTS = false, int CLN = 32>
__global__ void __launch_bounds__(256)
col_quant_kernel(const char *__restrict__ x, int8_t *__restrict__ q, void *__restrict__ scale,
                 const unsigned *__restrict__ amax_ws, int64_t P, int64_t Q, int64_t ld_bytes,
                 int64_t ldq, float eps, unsigned long long seed, unsigned long long offset, int nparts) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    constexpr int RL = 256 / CLN;
    const int cl = threadIdx.x % CLN, rl = threadIdx.x / CLN;
    const int64_t col0 = ((int64_t)blockIdx.x * CLN + cl) * VEC;
    const int64_t row0 = (int64_t)blockIdx.y * COL_PB;
    __shared__ unsigned cm[PARTS ? CLN * VEC : 1];
    // all of this thread's rows are loaded before the first is used: 16 x 16 B in flight per lane, and they are ISSUED before the absmax words
    // are waited for (round 3: eight dword loads of the absmax and the divisor set-up sat in front of them, a dependent ~1.5 us at the head of
    // every block).  Row of slot i: interleaved over the row-lanes (i * RL + rl), except under stochastic rounding, where a thread takes NIT
    // CONSECUTIVE rows: logical element indices of the (Q, P) view run along the rows, and four consecutive ones share a Philox block.
    constexpr int NIT = COL_PB / RL;
    auto row_of = [&](int i) -> int64_t { return SR ? row0 + rl * NIT + i : row0 + i * RL + rl; };
    v4i raw[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int64_t row = row_of(i);
        raw[i] = (v4i){0, 0, 0, 0};
        if (row < P && col0 < Q) raw[i] = *(const v4i *)(x + row * ld_bytes + col0 * (16 / VEC));
    }
    if constexpr (PARTS) {
        // the block's 32 * VEC columns, one per thread; the partial rows are L2 hits shared by every block of this column strip
        for (int c = threadIdx.x; c < CLN * VEC; c += 256) {
            const int64_t col = (int64_t)blockIdx.x * CLN * VEC + c;
            unsigned v = 0u;
            if (col < Q) {
                for (int g0 = 0; g0 < nparts; g0 += 8) {
                    unsigned w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = g0 + u < nparts ? amax_ws[(int64_t)(g0 + u) * Q + col] : 0u;
#pragma unroll
                    for (int u = 0; u < 8; ++u) v = w[u] > v ? w[u] : v;
                }
            }
            cm[c] = v;
        }
        __syncthreads();
    }
    if (col0 >= Q) return;
    unsigned am[VEC];
    if constexpr (PARTS) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) am[j] = cm[cl * VEC + j];
    } else if (((uintptr_t)amax_ws & 15) != 0) {   // a caller's absmax vector need only be 4-byte aligned (include/gnk.h)
#pragma unroll
        for (int j = 0; j < VEC; ++j) am[j] = amax_ws[col0 + j];
    } else {                                   // col0 is a multiple of VEC: VEC / 4 wide loads
#pragma unroll
        for (int j4 = 0; j4 < VEC / 4; ++j4) {
            const v4i w = *(const v4i *)(amax_ws + col0 + 4 * j4);
            am[4 * j4] = (unsigned)w[0]; am[4 * j4 + 1] = (unsigned)w[1]; am[4 * j4 + 2] = (unsigned)w[2]; am[4 * j4 + 3] = (unsigned)w[3];
        }
    }
    GnkDivisor dv[VEC];
    bool fast = true;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float s = __uint_as_float(am[j]) / 127.0f;
        dv[j].set(gnk_clamp_min_nan(s, eps));
        fast = fast && dv[j].ok();
        if (blockIdx.y == 0 && rl == 0) gnk_store_scalar_rt(scale, DT, col0 + j, s);
    }
    auto store_row = [&](int64_t row, const int (&qi)[VEC]) {
        if constexpr (VEC == 8) {
            v2i o = {(int)pack4_i8(qi[0], qi[1], qi[2], qi[3]), (int)pack4_i8(qi[4], qi[5], qi[6], qi[7])};
            if constexpr (NT) __builtin_nontemporal_store(o, (v2i *)(q + row * ldq + col0));
            else *(v2i *)(q + row * ldq + col0) = o;
        } else {
            *(unsigned *)(q + row * ldq + col0) = pack4_i8(qi[0], qi[1], qi[2], qi[3]);
        }
    };
    if constexpr (SR) {
        const bool aligned = ((P | (int64_t)offset) & 3) == 0;      // every 4-row group then starts a Philox block
#pragma unroll
        for (int g = 0; g < NIT / 4; ++g) {
            float f[4][VEC];
            int qi[4][VEC];
#pragma unroll
            for (int k = 0; k < 4; ++k) unpack16<DT>(raw[4 * g + k], f[k]);
            const int64_t r0 = row_of(4 * g);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const unsigned long long idx0 = (unsigned long long)((col0 + j) * P + r0);   // logical element (r = col, c = row) of the (Q, P) view
                float u[4];
                if (aligned) gnk_sr_uniform4(seed, offset, idx0, u);
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[k] = gnk_sr_uniform(seed, offset, idx0 + k);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) qi[k][j] = gnk_sr_clip_i8(fast ? dv[j].quot(f[k][j]) : f[k][j] / dv[j].d, u[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r0 + k < P) store_row(r0 + k, qi[k]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int64_t row = row_of(i);
            if (row < P) {
                float f[VEC];
                unpack16<DT>(raw[i], f);
                int qi[VEC];
                if (fast) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) qi[j] = gnk_rne_clip_i8(dv[j].quot(f[j]));
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) qi[j] = gnk_rne_clip_i8_any(f[j] / dv[j].d);
                }
                store_row(row, qi);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// generic: any strides, one block per logical row, two strided passes.
// ------------------------------------------------------------------------------------------
template <bool SR>
__global__ void __launch_bounds__(256)
quant_generic_kernel(const void *__restrict__ x, int dt, int8_t *__restrict__ q, void *__restrict__ scale,
                     int64_t R, int64_t C, int64_t sxr, int64_t sxc, int64_t sqr, int64_t sqc, float eps,
                     unsigned long long seed, unsigned long long offset) {
    __shared__ unsigned red[4];
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        unsigned mb = 0u;                                  // |x| maximum as an fp32 bit pattern (NaN-propagating, GnkAbsMax)
        for (int64_t c = threadIdx.x; c < C; c += 256) {
            const unsigned a = __float_as_uint(gnk_load_scalar_rt(x, dt, r * sxr + c * sxc)) & 0x7fffffffu;
            mb = a > mb ? a : mb;
        }
        mb = wave_umax(mb);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mb;
        __syncthreads();
        const unsigned m01 = red[0] > red[1] ? red[0] : red[1], m23 = red[2] > red[3] ? red[2] : red[3];
        const float amax = __uint_as_float(m01 > m23 ? m01 : m23);
        const float s = amax / 127.0f;
        const float div = gnk_clamp_min_nan(s, eps);
        if (threadIdx.x == 0) gnk_store_scalar_rt(scale, dt, r, s);
        for (int64_t c = threadIdx.x; c < C; c += 256)
            q[r * sqr + c * sqc] = (int8_t)gnk_quant_one<SR>(gnk_load_scalar_rt(x, dt, r * sxr + c * sxc), div, seed, offset,
                                                             (unsigned long long)(r * C + c));
    }
}

template <int DT, bool SR>
int launch_rowwise(const void *x, int8_t *q, void *scale, int64_t R, int64_t C, int64_t sxr, int64_t sqr, float eps,
                   uint64_t seed, uint64_t offset, hipStream_t st) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    const int64_t sxr_bytes = sxr * (16 / VEC);
    const int64_t per_lane = (C / VEC + 63) / 64;
    // at most 8 waves per CU: each wave then pipelines R / (8 * CUs) rows
    // one row per wave until the grid exceeds 64 blocks per CU; beyond that waves loop over rows
    const int64_t max_blocks = (int64_t)gnk_num_cus() * 64;
    const int64_t want = (R + 3) / 4;
    const dim3 grid((unsigned)(want < max_blocks ? want : max_blocks)), block(256);
    // Nontemporal (streaming) stores for the int8 copy: 12.1 -> 9.5 us at 4096^2 bf16 (the copy is read back by the GEMM
    // through L2 anyway).  Debug flag 16 restores plain stores for A/B timing.
    const bool nt = !(gnk_debug_flags_get() & 16);
#define GNK_ROW(NCH)                                                                                              \
    do {                                                                                                          \
        if (nt) hipLaunchKernelGGL((quant_rowwise_kernel<DT, NCH, SR, true>), grid, block, 0, st, (const char *)x, q, scale, R, C, sxr_bytes, sqr, eps, seed, offset); \
        else hipLaunchKernelGGL((quant_rowwise_kernel<DT, NCH, SR, false>), grid, block, 0, st, (const char *)x, q, scale, R, C, sxr_bytes, sqr, eps, seed, offset); \
    } while (0)
    if (per_lane <= 1) GNK_ROW(1);
    else if (per_lane <= 2) GNK_ROW(2);
    else if (per_lane <= 4) GNK_ROW(4);
    else if (per_lane <= 8) GNK_ROW(8);
    else if (per_lane <= 16) GNK_ROW(16);
    else
        hipLaunchKernelGGL((quant_rowwise_long_kernel<DT, SR>), dim3((unsigned)R), block, 0, st, (const char *)x, q, scale,
                           R, C, sxr_bytes, sqr, eps, seed, offset);
#undef GNK_ROW
    return gnk_check_launch("quantize_int8 rowwise");
}

template <int DT, bool SR>
int launch_colwise(const void *x, int8_t *q, void *scale, int64_t P, int64_t Q, int64_t ld, int64_t ldq, float eps,
                   uint64_t seed, uint64_t offset, unsigned *ws, int64_t ws_bytes, hipStream_t st) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    const int64_t ld_bytes = ld * (16 / VEC);
    const int64_t gx = (Q + 32 * VEC - 1) / (32 * VEC);
    const dim3 block(256);
    const bool nt = !(gnk_debug_flags_get() & 16);
    const bool big = gx * ((P + 127) / 128) * 2 >= (int64_t)gnk_num_cus();   // measured: 128-row slabs win down to 1 block per CU (4096^2, 8192 x 1024)
    const dim3 grid((unsigned)gx, (unsigned)(big ? (P + 127) / 128 : (P + 31) / 32));
    int64_t rpb, nparts;
    col_partial_plan(P, true, rpb, nparts);
    // col_absmax_kernel's geometry from one block per two CUs up (tools/colwise_ab.py: 1024 x 4096 11.3 vs 14.2 us, 5632 x 2048 14.0 vs 18.1 us;
    // 8192 x 1024, the other 128-block case, 14.1 vs 13.7 us)
    const bool wide = gx * nparts * 2 >= (int64_t)gnk_num_cus();
    if (!wide) col_partial_plan(P, false, rpb, nparts);
    if (ws_bytes >= nparts * Q * 4) {
        // partial maxima: nothing to clear, no atomics (the workspace gnk_quantize_int8_workspace_bytes asks for)
        if (wide) hipLaunchKernelGGL((col_absmax_partial_kernel<DT, 32>), dim3((unsigned)gx, (unsigned)nparts), block, 0, st, (const char *)x, ws, P, Q, ld_bytes, rpb);
        else hipLaunchKernelGGL((col_absmax_partial_kernel<DT, 8>), dim3((unsigned)((Q + 8 * VEC - 1) / (8 * VEC)), (unsigned)nparts), block, 0, st, (const char *)x, ws, P, Q, ld_bytes, rpb);
        if (!big) hipLaunchKernelGGL((col_quant_kernel<DT, SR, 32, false, true>), grid, block, 0, st, (const char *)x, q, scale, ws, P, Q, ld_bytes, ldq, eps, seed, offset, (int)nparts);
        else if (nt) hipLaunchKernelGGL((col_quant_kernel<DT, SR, 128, true, true>), grid, block, 0, st, (const char *)x, q, scale, ws, P, Q, ld_bytes, ldq, eps, seed, offset, (int)nparts);
        else hipLaunchKernelGGL((col_quant_kernel<DT, SR, 128, false, true>), grid, block, 0, st, (const char *)x, q, scale, ws, P, Q, ld_bytes, ldq, eps, seed, offset, (int)nparts);
        return gnk_check_launch("quantize_int8 colwise");
    }
    // a workspace of one word per logical row (the minimum; gnk_quantize_int8_noalloc beyond what its scratch holds as partial rows):
    // atomicMax into words cleared on the stream first
    if (hipMemsetAsync(ws, 0, (size_t)Q * 4, st) != hipSuccess) return gnk_fail(GNK_ERR_LAUNCH, "quantize_int8: memset of workspace failed");
    if (big) {
        hipLaunchKernelGGL((col_absmax_kernel<DT, 128>), grid, block, 0, st, (const char *)x, ws, P, Q, ld_bytes);
        if (nt) hipLaunchKernelGGL((col_quant_kernel<DT, SR, 128, true>), grid, block, 0, st, (const char *)x, q, scale, ws, P, Q, ld_bytes, ldq, eps, seed, offset, 0);
        else hipLaunchKernelGGL((col_quant_kernel<DT, SR, 128, false>), grid, block, 0, st, (const char *)x, q, scale, ws, P, Q, ld_bytes, ldq, eps, seed, offset, 0);
    } else {
        hipLaunchKernelGGL((col_absmax_kernel<DT, 32>), grid, block, 0, st, (const char *)x, ws, P, Q, ld_bytes);
        hipLaunchKernelGGL((col_quant_kernel<DT, SR, 32, false>), grid, block, 0, st, (const char *)x, q, scale, ws, P, Q, ld_bytes, ldq, eps, seed, offset, 0);
    }
    return gnk_check_launch("quantize_int8 colwise");
}

inline int64_t rowcol_blocks(int64_t R) {
    const int64_t want = (R + 7) / 8, cap = (int64_t)gnk_num_cus() * GNK_RC_CAP;
    return want < cap ? want : cap;
}

template <int DT, bool SR>
int launch_rowwise_colmax(const void *x, int8_t *q, void *scale, float *colmax, int64_t R, int64_t C, int64_t sxr, int64_t sqr, float eps,
                          uint64_t seed, uint64_t offset, unsigned *partial, hipStream_t st) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    const int64_t sxr_bytes = sxr * (16 / VEC);
    const int64_t per_lane = (C / VEC + 63) / 64;
    const int64_t G = rowcol_blocks(R);
    const dim3 grid((unsigned)G), block(512);
    const size_t lds = (size_t)C * 4;
#define GNK_ROWCOL(NCH) hipLaunchKernelGGL((quant_rowwise_colmax_kernel<DT, NCH, SR, true>), grid, block, lds, st, (const char *)x, q, scale, partial, R, C, sxr_bytes, sqr, eps, seed, offset)
    if (per_lane <= 1) GNK_ROWCOL(1);
    else if (per_lane <= 2) GNK_ROWCOL(2);
    else if (per_lane <= 4) GNK_ROWCOL(4);
    else if (per_lane <= 8) GNK_ROWCOL(8);
    else GNK_ROWCOL(12);   // row + prefetched row + column maxima = 3 x 4 x NCH registers: 12 chunks is what 256 VGPRs hold without spilling
#undef GNK_ROWCOL
    int rc = gnk_check_launch("quantize_int8 rowwise + column absmax");
    if (rc) return rc;
    hipLaunchKernelGGL(colmax_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, st, partial, (unsigned *)colmax, G, C);
    return gnk_check_launch("quantize_int8 column absmax reduction");
}

// pass 2 of the strided reduction alone: the column absmax is given (fp32, e.g. from launch_rowwise_colmax)
template <int DT, bool SR>
int launch_colwise_given(const void *x, int8_t *q, void *scale, int64_t P, int64_t Q, int64_t ld, int64_t ldq, float eps,
                         uint64_t seed, uint64_t offset, const unsigned *amax, hipStream_t st) {
    constexpr int VEC = ElemTraits<DT>::VEC;
    const int64_t ld_bytes = ld * (16 / VEC);
    const int64_t gx = (Q + 32 * VEC - 1) / (32 * VEC);
    const dim3 block(256);
    // measured (tools/quant_step_probe.py, cold 32768-row tensors): 2048 columns 48.6 vs 50.2 us, 5632 columns (three blocks per row) 116.6 vs 115.3
    if (GNK_COLQ_WIDE && Q >= 128 * VEC && Q <= 256 * VEC && ((P + 15) / 16) * ((Q + 256 * VEC - 1) / (256 * VEC)) * 2 >= (int64_t)gnk_num_cus()) {
        // wide matrices: one row-lane, 256 column-lanes, 16 rows per block
        const dim3 grid((unsigned)((Q + 256 * VEC - 1) / (256 * VEC)), (unsigned)((P + 15) / 16));
        hipLaunchKernelGGL((col_quant_kernel<DT, SR, 16, true, false, 256>), grid, block, 0, st, (const char *)x, q, scale, amax, P, Q, ld_bytes, ldq, eps, seed, offset, 0);
    } else if (gx * ((P + 127) / 128) * 2 >= (int64_t)gnk_num_cus()) {
        const dim3 grid((unsigned)gx, (unsigned)((P + GNK_COLQ_PB - 1) / GNK_COLQ_PB));
        hipLaunchKernelGGL((col_quant_kernel<DT, SR, GNK_COLQ_PB, true>), grid, block, 0, st, (const char *)x, q, scale, amax, P, Q, ld_bytes, ldq, eps, seed, offset, 0);
    } else {
        const dim3 grid((unsigned)gx, (unsigned)((P + 31) / 32));
        hipLaunchKernelGGL((col_quant_kernel<DT, SR, 32, false>), grid, block, 0, st, (const char *)x, q, scale, amax, P, Q, ld_bytes, ldq, eps, seed, offset, 0);
    }
    return gnk_check_launch("quantize_int8 colwise (given absmax)");
}

inline bool aligned(const void *p, int a) { return ((uintptr_t)p % a) == 0; }

// the strided-reduction kernels (the only ones that need the workspace) take the call: shared by gnk_quantize_int8 and its no-alloc form
inline bool colwise_fast_path(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int64_t sxc, const void *q, int64_t sqr, int64_t sqc) {
    const int esz = gnk_dtype_size(x_dtype), vec = 16 / esz;
    const bool rowwise = C > 0 && sxc == 1 && sqc == 1 && C % vec == 0 && aligned(x, 16) && (sxr * esz) % 16 == 0 && aligned(q, vec) &&
                         sqr % vec == 0 && sxr >= C && sqr >= C;
    return !rowwise && C > 0 && sxr == 1 && sqr == 1 && R % vec == 0 && aligned(x, 16) && (sxc * esz) % 16 == 0 && aligned(q, vec) &&
           sqc % vec == 0 && sxc >= R && sqc >= R;
}

}  // namespace

// Strided reduction: the workspace asked for holds the partial maxima of pass 1 (<= GNK_COL_PARTS rows of R words: nothing to clear);
// gnk_quantize_int8 also accepts the minimum of R * 4 bytes and then clears + atomicMax-es one word per logical row.
extern "C" int64_t gnk_quantize_int8_workspace_bytes(int64_t R, int64_t C, int64_t sxr, int64_t sxc) {
    if (sxc == 1 || R <= 0 || sxr != 1) return 0;
    int64_t rpb, na, nb;                                  // enough for either pass-1 geometry (the choice depends on the dtype too)
    col_partial_plan(C > 0 ? C : 1, true, rpb, na);
    col_partial_plan(C > 0 ? C : 1, false, rpb, nb);
    return (na > nb ? na : nb) * R * 4;
}

extern "C" int gnk_quantize_int8(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int64_t sxc,
                                 int8_t *q, int64_t sqr, int64_t sqc, void *scale, float eps, int sr, uint64_t seed,
                                 uint64_t offset, void *workspace, int64_t workspace_bytes, void *stream) {
    if (R < 0 || C < 0) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8: negative size (%lld, %lld)", (long long)R, (long long)C);
    if (!gnk_is_float_dtype(x_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8: input dtype code %d is not bf16/f16/f32", x_dtype);
    if (R == 0) return GNK_OK;
    if (!scale) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8: null scale pointer");
    if (C > 0 && (!x || !q)) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8: null data pointer");
    int rc = gnk_init();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int esz = gnk_dtype_size(x_dtype);
    const int vec = 16 / esz;

#define GNK_DISPATCH(FN, ...)                                                         \
    do {                                                                              \
        if (x_dtype == GNK_BF16) return sr ? FN<GNK_BF16, true>(__VA_ARGS__) : FN<GNK_BF16, false>(__VA_ARGS__); \
        if (x_dtype == GNK_F16) return sr ? FN<GNK_F16, true>(__VA_ARGS__) : FN<GNK_F16, false>(__VA_ARGS__);    \
        return sr ? FN<GNK_F32, true>(__VA_ARGS__) : FN<GNK_F32, false>(__VA_ARGS__); \
    } while (0)

    // reduction along the contiguous dim
    if (C > 0 && sxc == 1 && sqc == 1 && C % vec == 0 && aligned(x, 16) && (sxr * esz) % 16 == 0 && aligned(q, vec) &&
        sqr % vec == 0 && sxr >= C && sqr >= C) {
        GNK_DISPATCH(launch_rowwise, x, q, scale, R, C, sxr, sqr, eps, seed, offset, st);
    }
    // reduction along the strided dim: memory is [P = C][Q = R] with Q contiguous
    if (colwise_fast_path(x, x_dtype, R, C, sxr, sxc, q, sqr, sqc)) {
        if (!workspace || workspace_bytes < R * 4)
            return gnk_fail(GNK_ERR_WORKSPACE, "quantize_int8: strided reduction needs a workspace of %lld bytes", (long long)(R * 4));
        GNK_DISPATCH(launch_colwise, x, q, scale, /*P=*/C, /*Q=*/R, /*ld=*/sxc, /*ldq=*/sqc, eps, seed, offset,
                     (unsigned *)workspace, workspace_bytes, st);
    }
#undef GNK_DISPATCH
    // anything else (C == 0 included: scale = 0, nothing to write)
    const unsigned grid = (unsigned)(R < 4096 ? R : 4096);
    if (sr)
        hipLaunchKernelGGL((quant_generic_kernel<true>), dim3(grid), dim3(256), 0, st, x, x_dtype, q, scale, R, C, sxr, sxc,
                           sqr, sqc, eps, seed, offset);
    else
        hipLaunchKernelGGL((quant_generic_kernel<false>), dim3(grid), dim3(256), 0, st, x, x_dtype, q, scale, R, C, sxr, sxc,
                           sqr, sqc, eps, seed, offset);
    return gnk_check_launch("quantize_int8 generic");
}

// ---- the entry point that allocates nothing (include/gnk.h: gnk_quantize_int8_noalloc) ---------------------------
// The strided reduction needs one fp32 absmax word per logical row between its two kernels.  gnk_quantize_int8 takes
// that buffer from the caller; here it lives in the code object: every GPU that loads the library holds its own copy
// of `scratch` (GNK_NOALLOC_SCRATCH_BYTES, until the process ends) at its own address.  In host code the array's name
// is only a handle: the device address comes from hipGetSymbolAddress, once per device, and is what the kernels and
// hipMemsetAsync receive.
static __device__ __align__(256) unsigned char scratch[GNK_NOALLOC_SCRATCH_BYTES];
static void *g_scratch_ptr[64];
static std::mutex g_scratch_mu;

// Called by gnk_init() (gnk_api.hip) with the device current; idempotent.
int gnk_quantize_noalloc_setup(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return gnk_fail(GNK_ERR_LAUNCH, "quantize_int8_noalloc: cannot identify the current device");
    std::lock_guard<std::mutex> guard(g_scratch_mu);
    if (g_scratch_ptr[dev]) return GNK_OK;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
        return gnk_fail(GNK_ERR_LAUNCH, "quantize_int8_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev] = ptr;
    return GNK_OK;
}

extern "C" int64_t gnk_quantize_int8_noalloc_max_rows(void) { return GNK_NOALLOC_SCRATCH_BYTES / 4; }

extern "C" int gnk_quantize_int8_noalloc(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int64_t sxc,
                                         int8_t *q, int64_t sqr, int64_t sqc, void *scale, float eps, int sr, uint64_t seed,
                                         uint64_t offset, void *stream) {
    // only the strided-reduction kernels use the scratch (one word per logical row); every other input goes through untouched
    const int64_t need = gnk_is_float_dtype(x_dtype) && colwise_fast_path(x, x_dtype, R, C, sxr, sxc, q, sqr, sqc) ? R * 4 : 0;
    if (need > (int64_t)GNK_NOALLOC_SCRATCH_BYTES)
        return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_int8_noalloc: a strided reduction over %lld logical rows needs %lld bytes of scratch, "
                        "the built-in buffer holds %lld (at most %lld rows): use gnk_quantize_int8 with a caller workspace",
                        (long long)R, (long long)need, (long long)GNK_NOALLOC_SCRATCH_BYTES, (long long)(GNK_NOALLOC_SCRATCH_BYTES / 4));
    int rc = gnk_init();              // runs gnk_quantize_noalloc_setup() for the current device (no allocation: a symbol look-up)
    if (rc) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_scratch_ptr[dev])
        return gnk_fail(GNK_ERR_LAUNCH, "quantize_int8_noalloc: no scratch address for the current device");
    void *ptr = g_scratch_ptr[dev];   // the DEVICE address; never the array's name
    // launch_colwise either writes every partial word it later reads (small reductions) or clears the words it uses (hipMemsetAsync on
    // `stream`) before pass 1, so the previous call's maxima do not leak
    return gnk_quantize_int8(x, x_dtype, R, C, sxr, sxc, q, sqr, sqc, scale, eps, sr, seed, offset, ptr, GNK_NOALLOC_SCRATCH_BYTES, stream);
}

// ---- fused dual-direction entry points (include/gnk.h) ---------------------------------------------------------
static bool rowcol_supported(int x_dtype, int64_t R, int64_t C) {
    if (!gnk_is_float_dtype(x_dtype) || R <= 0 || C <= 0) return false;
    const int vec = 16 / gnk_dtype_size(x_dtype);
    return C % vec == 0 && (C / vec + 63) / 64 <= 12;   // rows of up to 6144 16-bit / 3072 fp32 elements; longer rows take the two separate passes
}

extern "C" int64_t gnk_quantize_int8_rowwise_colmax_workspace_bytes(int x_dtype, int64_t R, int64_t C) {
    if (!rowcol_supported(x_dtype, R, C)) return 0;
    return rowcol_blocks(R) * C * 4;
}

extern "C" int gnk_quantize_int8_rowwise_colmax(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int8_t *q, int64_t sqr,
                                                void *scale, float *colmax, float eps, int sr, uint64_t seed, uint64_t offset,
                                                void *workspace, int64_t workspace_bytes, void *stream) {
    if (!rowcol_supported(x_dtype, R, C)) return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_int8_rowwise_colmax: shape (%lld, %lld) / dtype code %d has no fused kernel", (long long)R, (long long)C, x_dtype);
    if (!x || !q || !scale || !colmax) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8_rowwise_colmax: null pointer");
    const int esz = gnk_dtype_size(x_dtype), vec = 16 / esz;
    if (!(aligned(x, 16) && (sxr * esz) % 16 == 0 && aligned(q, vec) && sqr % vec == 0 && sxr >= C && sqr >= C && aligned(colmax, 4)))
        return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_int8_rowwise_colmax: rows must be contiguous, 16-byte aligned");
    int rc = gnk_init();
    if (rc) return rc;
    const int64_t need = rowcol_blocks(R) * C * 4;
    if (!workspace || workspace_bytes < need || !aligned(workspace, 16))
        return gnk_fail(GNK_ERR_WORKSPACE, "quantize_int8_rowwise_colmax: needs a 16-byte aligned workspace of %lld bytes", (long long)need);
    hipStream_t st = (hipStream_t)stream;
#define GNK_DISPATCH2(FN, ...)                                                         \
    do {                                                                              \
        if (x_dtype == GNK_BF16) return sr ? FN<GNK_BF16, true>(__VA_ARGS__) : FN<GNK_BF16, false>(__VA_ARGS__); \
        if (x_dtype == GNK_F16) return sr ? FN<GNK_F16, true>(__VA_ARGS__) : FN<GNK_F16, false>(__VA_ARGS__);    \
        return sr ? FN<GNK_F32, true>(__VA_ARGS__) : FN<GNK_F32, false>(__VA_ARGS__); \
    } while (0)
    GNK_DISPATCH2(launch_rowwise_colmax, x, q, scale, colmax, R, C, sxr, sqr, eps, seed, offset, (unsigned *)workspace, st);
}

extern "C" int gnk_quantize_int8_colwise_given_absmax(const void *x, int x_dtype, int64_t P, int64_t Q, int64_t ld, int8_t *q, int64_t ldq,
                                                      void *scale, const float *absmax, float eps, int sr, uint64_t seed, uint64_t offset,
                                                      void *stream) {
    if (!gnk_is_float_dtype(x_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8_colwise_given_absmax: input dtype code %d is not bf16/f16/f32", x_dtype);
    if (P < 0 || Q < 0) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8_colwise_given_absmax: negative size");
    if (P == 0 || Q == 0) return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_int8_colwise_given_absmax: empty input");
    if (!x || !q || !scale || !absmax) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_int8_colwise_given_absmax: null pointer");
    const int esz = gnk_dtype_size(x_dtype), vec = 16 / esz;
    if (!(Q % vec == 0 && aligned(x, 16) && (ld * esz) % 16 == 0 && aligned(q, vec) && ldq % vec == 0 && ld >= Q && ldq >= Q))
        return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_int8_colwise_given_absmax: rows must be contiguous, 16-byte aligned, Q a multiple of %d", vec);
    int rc = gnk_init();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    GNK_DISPATCH2(launch_colwise_given, x, q, scale, P, Q, ld, ldq, eps, seed, offset, (const unsigned *)absmax, st);
#undef GNK_DISPATCH2
}
