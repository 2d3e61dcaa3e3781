// gnk_quantize_mx.hip -- OCP microscaling (MX) quantization for gfx950: one E8M0 scale per 32 consecutive elements of a
// row, data as FP4 e2m1 (two per byte), FP8 e4m3fn or FP8 e5m2.  HBM-bound, one pass.
//
// Restates kernels/gn_kernels/utils.py:83-107 (quantize_mx) of the reference with its two scale rules
// (:30-41 "nv", :45-49 "ocp") and its FP4 rounding table (:52-80), bit for bit:
//   amax  = max |x| over the block (fp32)
//   bits  = ocp: max(exponent_field(amax) - log2(pow2_amax), 0)      pow2_amax = 4 / 256 / 32768
//           nv : exponent_field(amax / dtype_amax) rounded up when the quotient is not a power of two
//   y     = clip(x / max(as_float(bits << 23), 1e-12), +-dtype_amax)  true fp32 division (the 1e-12 floor is not a power of 2)
//   data  = FP4: nearest e2m1 with ties to even mantissa (the <= / < chain of :59-65); FP8: round-to-nearest-even cast
// Lanes: 32 / VEC adjacent lanes share a block (VEC = elements per 16-byte load: 8 for bf16 / f16, 4 for f32), so loads are
// 16 B per lane and fully coalesced, the block absmax is 2-3 shuffles, and the data goes out as 4 / 8 bytes per lane.
#include "gnk_common.h"

namespace {

enum { MX_FP4 = 7 };   // q_format codes: GNK_E4M3, GNK_E5M2 (include/gnk.h) and 7 = FP4 e2m1 x2

__device__ __forceinline__ unsigned fp4_code(float y) {
    const float a = fabsf(y);
    unsigned c = a <= 5.0f ? 6u : 7u;
    c = a < 3.5f ? 5u : c;
    c = a <= 2.5f ? 4u : c;
    c = a < 1.75f ? 3u : c;
    c = a <= 1.25f ? 2u : c;
    c = a < 0.75f ? 1u : c;
    c = a <= 0.25f ? 0u : c;
    return ((__float_as_uint(y) >> 31) << 3) | c;
}

// fp32 -> fp8 code, round to nearest even, the value is already clipped to the format's finite range.  EB / MB: exponent /
// mantissa bits (4, 3: e4m3fn bias 7; 5, 2: e5m2 bias 15).  Integer arithmetic so that ties and subnormals are exact.
template <int EB, int MB> __device__ __forceinline__ unsigned fp8_code(float y) {
    constexpr int BIAS = (1 << (EB - 1)) - 1;
    const unsigned u = __float_as_uint(y), sign = (u >> 31) << 7, mag = u & 0x7fffffffu;
    const int e = (int)(mag >> 23) - 127;                       // unbiased fp32 exponent
    unsigned code;
    if (mag == 0u) code = 0u;
    else if (e >= 1 - BIAS) {                                   // normal in the target format
        const unsigned m = mag & 0x7fffffu, drop = 23 - MB;
        unsigned q = ((unsigned)(e + BIAS) << MB) | (m >> drop);
        const unsigned rem = m & ((1u << drop) - 1u), half = 1u << (drop - 1);
        q += (rem > half || (rem == half && (q & 1u))) ? 1u : 0u;   // carry into the exponent is the correct next binade
        code = q;
    } else {                                                    // subnormal: value = k * 2^(1 - BIAS - MB)
        const float scaled = y * __uint_as_float((unsigned)(127 + BIAS - 1 + MB) << 23);   // exact: power-of-two multiply
        code = (unsigned)(int)rintf(fabsf(scaled));             // half-to-even; k == 2^MB lands on the smallest normal
    }
    return sign | code;
}

typedef short v2s __attribute__((ext_vector_type(2)));
template <int SEL> __device__ __forceinline__ unsigned hw_fp4(unsigned old, float a, float b) { return __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(old, a, b, 1.0f, SEL); }
template <int FMT, bool HI> __device__ __forceinline__ v2s hw_fp8(v2s old, float a, float b) {
    if constexpr (FMT == GNK_E4M3) return __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 1.0f, HI);
    else return __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(old, a, b, 1.0f, HI);
}

template <int DT> struct MxElem;
template <> struct MxElem<GNK_BF16> { static constexpr int VEC = 8; };
template <> struct MxElem<GNK_F16>  { static constexpr int VEC = 8; };
template <> struct MxElem<GNK_F32>  { static constexpr int VEC = 4; };

constexpr int MX_UNR = 4;   // 16-byte chunks per thread, all loaded before the first is used (bytes in flight: one chunk per thread
                            // kept 32 KiB per CU in flight, half of what the HBM latency needs)

template <int DT, int FMT, int METHOD>
__global__ void __launch_bounds__(256) quantize_mx_kernel(const char *__restrict__ x, unsigned char *__restrict__ q, unsigned char *__restrict__ scales,
                                                          int64_t M, int64_t N, int64_t ldx_bytes, int64_t ldq, int64_t lds) {
    constexpr int VEC = MxElem<DT>::VEC, LPB = 32 / VEC;       // lanes per MX block
    const int64_t chunks_per_row = N / VEC;
    const int64_t total = M * chunks_per_row;
    v4i raws[MX_UNR];
    int64_t rows[MX_UNR], chs[MX_UNR];
    bool lives[MX_UNR];
#pragma unroll
    for (int u = 0; u < MX_UNR; ++u) {
        const int64_t gid = ((int64_t)blockIdx.x * MX_UNR + u) * 256 + threadIdx.x;
        lives[u] = gid < total;                                 // whole blocks are live or dead together (N % 32 == 0)
        // 32-bit division (the host guarantees total < 2^32): a 64-bit one per chunk was most of this kernel's time
        const unsigned r32 = lives[u] ? (unsigned)gid / (unsigned)chunks_per_row : 0u;
        rows[u] = r32;
        chs[u] = lives[u] ? (int64_t)((unsigned)gid - r32 * (unsigned)chunks_per_row) : 0;
        raws[u] = (v4i){0, 0, 0, 0};
        if (lives[u]) raws[u] = *(const v4i *)(x + rows[u] * ldx_bytes + chs[u] * 16);
    }
#pragma unroll
    for (int u = 0; u < MX_UNR; ++u) {
    const v4i raw = raws[u];
    const bool live = lives[u];
    const int64_t row = rows[u], ch = chs[u];
    float f[VEC];
    if constexpr (DT == GNK_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __int_as_float(raw[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned w = (unsigned)raw[i];
            if constexpr (DT == GNK_BF16) { f[2 * i] = __uint_as_float(w << 16); f[2 * i + 1] = __uint_as_float(w & 0xffff0000u); }
            else { f[2 * i] = gnk_f16_to_f32((unsigned short)(w & 0xffffu)); f[2 * i + 1] = gnk_f16_to_f32((unsigned short)(w >> 16)); }
        }
    }
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) amax = fmaxf(amax, fabsf(f[i]));
#pragma unroll
    for (int off = 1; off < LPB; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));

    constexpr float DT_AMAX = FMT == MX_FP4 ? 6.0f : FMT == GNK_E4M3 ? 448.0f : 57344.0f;
    constexpr int POW2_LOG = FMT == MX_FP4 ? 2 : FMT == GNK_E4M3 ? 8 : 15;
    int bits;
    if constexpr (METHOD == 0) {       // OCP: floor(log2 amax) - log2(pow2_amax) in the exponent field; below the smallest normal -> 0
        const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
        bits = e - POW2_LOG > 0 ? e - POW2_LOG : 0;
    } else {                           // NV: ceil to a power of two of amax / dtype_amax
        const unsigned sb = __float_as_uint(__fdiv_rn(amax, DT_AMAX));
        const int e = (int)(sb >> 23), m = (int)(sb & 0x7fffffu);
        const bool up = (e > 0 && e < 254 && m > 0) || (e == 0 && m > 0x400000);
        bits = up ? e + 1 : e;
    }
    const float div = fmaxf(__uint_as_float((unsigned)bits << 23), 1e-12f);
    // bits in [88, 253]: the divisor is 2^(bits - 127) >= 2^-39 > 1e-12 and its reciprocal 2^(127 - bits) is a normal fp32 number, so
    // x * reciprocal IS x / divisor (the exact quotient, rounded once either way) at 1 VALU instead of ~10.  Wave-uniform choice.
    const bool pow2 = __all(!live || (bits >= 88 && bits <= 253));
    const float rcp = __uint_as_float((unsigned)(254 - bits) << 23);
    auto quot = [&](float v) { return pow2 ? __fmul_rn(v, rcp) : __fdiv_rn(v, div); };
    // Rounding: the reference's FP4 table and its fp8 casts are round-to-nearest-even onto the target grid, which is what
    // gfx950's v_cvt_scalef32_pk_{fp4,fp8,bf8}_f32 do (scale operand 1.0: the division and the clip stay explicit, the
    // clip because amax / scale may exceed the format maximum under the "ocp" rule).  The integer restatements above remain
    // the path for blocks whose divisor is not a power of two, and the tests hold both to the same goldens.
    if constexpr (FMT == MX_FP4) {
        unsigned word = 0u;
        if (pow2) {
            float y[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) y[i] = fminf(fmaxf(__fmul_rn(f[i], rcp), -DT_AMAX), DT_AMAX);
            word = hw_fp4<0>(word, y[0], y[1]);
            word = hw_fp4<1>(word, y[2], y[3]);
            if constexpr (VEC == 8) {
                word = hw_fp4<2>(word, y[4], y[5]);
                word = hw_fp4<3>(word, y[6], y[7]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float y = quot(f[i]);
                y = fminf(fmaxf(y, -DT_AMAX), DT_AMAX);
                word |= fp4_code(y) << (4 * i);
            }
        }
        if (live) {
            if constexpr (VEC == 8) *(unsigned *)(q + row * ldq + ch * 4) = word;
            else *(unsigned short *)(q + row * ldq + ch * 2) = (unsigned short)word;
        }
    } else {
        unsigned w[2] = {0u, 0u};
        if (pow2) {
            float y[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) y[i] = fminf(fmaxf(__fmul_rn(f[i], rcp), -DT_AMAX), DT_AMAX);
#pragma unroll
            for (int h = 0; h < VEC / 4; ++h) {
                v2s pk = {0, 0};
                pk = hw_fp8<FMT, false>(pk, y[4 * h], y[4 * h + 1]);
                pk = hw_fp8<FMT, true>(pk, y[4 * h + 2], y[4 * h + 3]);
                w[h] = __builtin_bit_cast(unsigned, pk);
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float y = quot(f[i]);
                y = fminf(fmaxf(y, -DT_AMAX), DT_AMAX);
                const unsigned c = FMT == GNK_E4M3 ? fp8_code<4, 3>(y) : fp8_code<5, 2>(y);
                w[i >> 2] |= c << (8 * (i & 3));
            }
        }
        if (live) {
            if constexpr (VEC == 8) *(v2i *)(q + row * ldq + ch * 8) = (v2i){(int)w[0], (int)w[1]};
            else *(unsigned *)(q + row * ldq + ch * 4) = w[0];
        }
    }
    if (live && (ch % LPB) == 0) scales[row * lds + ch / LPB] = (unsigned char)bits;
    }
}

template <int DT, int FMT>
int launch_mx(int method, const void *x, void *q, void *scales, int64_t M, int64_t N, int64_t ldx, int64_t ldq, int64_t lds, hipStream_t st) {
    constexpr int VEC = MxElem<DT>::VEC;
    const int64_t total = M * (N / VEC);
    const dim3 grid((unsigned)((total + 256 * MX_UNR - 1) / (256 * MX_UNR))), block(256);
    const int64_t ldx_bytes = ldx * (16 / VEC);
    if (method == 0)
        hipLaunchKernelGGL((quantize_mx_kernel<DT, FMT, 0>), grid, block, 0, st, (const char *)x, (unsigned char *)q, (unsigned char *)scales, M, N, ldx_bytes, ldq, lds);
    else
        hipLaunchKernelGGL((quantize_mx_kernel<DT, FMT, 1>), grid, block, 0, st, (const char *)x, (unsigned char *)q, (unsigned char *)scales, M, N, ldx_bytes, ldq, lds);
    return gnk_check_launch("quantize_mx");
}

template <int DT>
int launch_mx_fmt(int fmt, int method, const void *x, void *q, void *scales, int64_t M, int64_t N, int64_t ldx, int64_t ldq, int64_t lds, hipStream_t st) {
    if (fmt == MX_FP4) return launch_mx<DT, MX_FP4>(method, x, q, scales, M, N, ldx, ldq, lds, st);
    if (fmt == GNK_E4M3) return launch_mx<DT, GNK_E4M3>(method, x, q, scales, M, N, ldx, ldq, lds, st);
    return launch_mx<DT, GNK_E5M2>(method, x, q, scales, M, N, ldx, ldq, lds, st);
}

}  // namespace

extern "C" int gnk_quantize_mx(const void *x, int x_dtype, int64_t M, int64_t N, int64_t ldx, void *q, int q_format, int64_t ldq,
                               void *scales, int64_t lds, int method, void *stream) {
    if (M < 0 || N < 0) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: negative size");
    if (!gnk_is_float_dtype(x_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: input dtype code %d is not bf16/f16/f32", x_dtype);
    if (q_format != MX_FP4 && q_format != GNK_E4M3 && q_format != GNK_E5M2) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: format code %d is not FP4 (7) / E4M3 / E5M2", q_format);
    if (method != 0 && method != 1) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: scale method %d is not 0 (ocp) / 1 (nv)", method);
    if (N % 32 != 0) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: row length %lld is not a multiple of the MX block (32)", (long long)N);
    if (M == 0 || N == 0) return GNK_OK;
    if (!x || !q || !scales) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: null pointer");
    const int esz = gnk_dtype_size(x_dtype);
    const int64_t qrow = q_format == MX_FP4 ? N / 2 : N;
    if (ldx < N || ldq < qrow || lds < N / 32) return gnk_fail(GNK_ERR_BAD_ARG, "quantize_mx: a leading dimension is shorter than its row");
    if (((uintptr_t)x & 15) || (ldx * esz) % 16 || ((uintptr_t)q & 7) || ldq % 8)
        return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_mx: rows of x must be 16-byte aligned, rows of q 8-byte aligned");
    if (M * (N / 4) >= (1LL << 32)) return gnk_fail(GNK_ERR_UNSUPPORTED, "quantize_mx: tensor too large for one launch (2^34 elements)");
    int rc = gnk_init();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == GNK_BF16) return launch_mx_fmt<GNK_BF16>(q_format, method, x, q, scales, M, N, ldx, ldq, lds, st);
    if (x_dtype == GNK_F16) return launch_mx_fmt<GNK_F16>(q_format, method, x, q, scales, M, N, ldx, ldq, lds, st);
    return launch_mx_fmt<GNK_F32>(q_format, method, x, q, scales, M, N, ldx, ldq, lds, st);
}
