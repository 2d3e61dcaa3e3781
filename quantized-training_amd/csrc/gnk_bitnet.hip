// gnk_bitnet.hip -- the BitNet b1.58 weight path of subclasses/bitnet.py for gfx950: tensor-wise abs-mean scale, ternary
// quantization (optionally straight into the 2-bit packed all-gather format), 2-bit pack / unpack.  All HBM-bound,
// contiguous tensors walked as flat 16-byte chunks (one chunk per lane and load, fully coalesced).
//
// Restates, bit for bit except for the summation order of the mean:
//   get_bitnet_scale       subclasses/bitnet.py:110-112   mean(|f32(x)|) over the whole tensor, fp32
//   quantize_bitnet_weight subclasses/bitnet.py:115-118   clip(round(f32(w) / max(scale, eps)), -1, 1) -> int8 (true division, half-even)
//   _pack_i2_in_i8         subclasses/bitnet.py:185-195   [a, b, c, d] -> (a << 6) | ((b & 3) << 4) | ((c & 3) << 2) | (d & 3)
//   _unpack_i2_in_i8       subclasses/bitnet.py:198-201   byte -> [x >> 6, x << 2 >> 6, x << 4 >> 6, x << 6 >> 6] (int8 arithmetic: sign extension)
// The mean is a two-level sum: fp32 within a 16-byte chunk (4 or 8 terms), fp64 from there on (per lane, across lanes and workgroups), in a fixed
// order (no atomics: same input -> same bits).  torch's own reduction order differs between devices and versions, so the
// scale is compared at 1e-6 relative (tests), everything downstream of a GIVEN scale bit-exactly.
#include "gnk_common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_UNR = 4;            // independent 16-byte chunks per lane and trip
constexpr int BN_MAX_PARTIALS = 1024;

template <int DT> struct BnTraits { static constexpr int VEC = 8; };
template <> struct BnTraits<GNK_F32> { static constexpr int VEC = 4; };

template <int DT> __device__ __forceinline__ void bn_unpack(const v4i raw, float *f) {
    if constexpr (DT == GNK_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __int_as_float(raw[j]);
    } else if constexpr (DT == GNK_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = __uint_as_float((unsigned)raw[j] << 16);
            f[2 * j + 1] = __uint_as_float((unsigned)raw[j] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = gnk_f16_to_f32((unsigned short)((unsigned)raw[j] & 0xffffu));
            f[2 * j + 1] = gnk_f16_to_f32((unsigned short)((unsigned)raw[j] >> 16));
        }
    }
}

__device__ __forceinline__ double bn_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide fp64 sum in a fixed order; every thread returns the total
__device__ __forceinline__ double bn_block_sum(double v, double *lds) {
    v = bn_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < BN_THREADS / 64; ++w) t += lds[w];
    return t;
}

// pass 1: one fp64 partial sum of |x| per workgroup (grid-stride over trips of BN_THREADS * BN_UNR chunks)
template <int DT>
__global__ void __launch_bounds__(BN_THREADS)
absmean_partial_kernel(const char *__restrict__ x, double *__restrict__ partials, int64_t numel) {
    constexpr int VEC = BnTraits<DT>::VEC;
    __shared__ double lds[BN_THREADS / 64];
    const int64_t nchunks = numel / VEC;
    const int64_t trip = (int64_t)BN_THREADS * BN_UNR;
    double acc[BN_UNR] = {};             // fp32 inside a 16-byte chunk, fp64 from there on: tensors of any size keep 1e-6
    for (int64_t base = (int64_t)blockIdx.x * trip; base < nchunks; base += (int64_t)gridDim.x * trip) {
        v4i raw[BN_UNR];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const int64_t c = base + u * BN_THREADS + threadIdx.x;
            raw[u] = c < nchunks ? *(const v4i *)(x + c * 16) : (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            float f[VEC];
            bn_unpack<DT>(raw[u], f);
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += fabsf(f[j]);
            acc[u] += (double)s;
        }
    }
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < BN_UNR; ++u) v += acc[u];
    if (blockIdx.x == 0 && threadIdx.x == 0)   // the last numel % VEC elements
        for (int64_t i = nchunks * VEC; i < numel; ++i) v += (double)fabsf(gnk_load_scalar<DT>(x, i));
    v = bn_block_sum(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = v;
}

// pass 2: sum of the partials / numel -> fp32 scalar
__global__ void __launch_bounds__(BN_THREADS)
absmean_final_kernel(const double *__restrict__ partials, int n, int64_t numel, float *__restrict__ out) {
    __shared__ double lds[BN_THREADS / 64];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += BN_THREADS) v += partials[i];
    v = bn_block_sum(v, lds);
    if (threadIdx.x == 0) out[0] = (float)(v / (double)numel);
}

__device__ __forceinline__ int bn_ternary(float x, const GnkDivisor dv, bool fast) {
    const float v = rintf(fast ? dv.quot(x) : x / dv.d);
    return (int)fminf(fmaxf(v, -1.0f), 1.0f);
}

__device__ __forceinline__ unsigned bn_pack4_i8(int a, int b, int c, int d) {
    return ((unsigned)a & 0xffu) | (((unsigned)b & 0xffu) << 8) | (((unsigned)c & 0xffu) << 16) | ((unsigned)d << 24);
}
__device__ __forceinline__ unsigned bn_pack4_i2(int a, int b, int c, int d) {
    return (((unsigned)a << 6) & 0xffu) | (((unsigned)b & 3u) << 4) | (((unsigned)c & 3u) << 2) | ((unsigned)d & 3u);
}

// ternary quantization given the scale; PACK: four consecutive values per output byte
template <int DT, bool PACK>
__global__ void __launch_bounds__(BN_THREADS)
bitnet_quant_kernel(const char *__restrict__ w, const float *__restrict__ scale, float eps, unsigned char *__restrict__ out, int64_t numel) {
    constexpr int VEC = BnTraits<DT>::VEC;
    const int64_t nchunks = numel / VEC;
    GnkDivisor dv;
    dv.set(fmaxf(scale[0], eps));
    const bool fast = dv.ok();
    const int64_t base = (int64_t)blockIdx.x * (BN_THREADS * BN_UNR);
    v4i raw[BN_UNR];
#pragma unroll
    for (int u = 0; u < BN_UNR; ++u) {
        const int64_t c = base + u * BN_THREADS + threadIdx.x;
        raw[u] = c < nchunks ? *(const v4i *)(w + c * 16) : (v4i){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < BN_UNR; ++u) {
        const int64_t c = base + u * BN_THREADS + threadIdx.x;
        if (c >= nchunks) continue;
        float f[VEC];
        bn_unpack<DT>(raw[u], f);
        int q[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) q[j] = bn_ternary(f[j], dv, fast);
        if constexpr (PACK) {
            if constexpr (VEC == 8) *(unsigned short *)(out + c * 2) = (unsigned short)(bn_pack4_i2(q[0], q[1], q[2], q[3]) | (bn_pack4_i2(q[4], q[5], q[6], q[7]) << 8));
            else out[c] = (unsigned char)bn_pack4_i2(q[0], q[1], q[2], q[3]);
        } else {
            if constexpr (VEC == 8) *(v2i *)(out + c * 8) = (v2i){(int)bn_pack4_i8(q[0], q[1], q[2], q[3]), (int)bn_pack4_i8(q[4], q[5], q[6], q[7])};
            else *(unsigned *)(out + c * 4) = bn_pack4_i8(q[0], q[1], q[2], q[3]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the last numel % VEC elements (a multiple of 4 when PACK)
        int64_t i = nchunks * VEC;
        if constexpr (PACK) {
            for (; i + 3 < numel; i += 4) {
                int q[4];
                for (int j = 0; j < 4; ++j) q[j] = bn_ternary(gnk_load_scalar<DT>(w, i + j), dv, fast);
                out[i / 4] = (unsigned char)bn_pack4_i2(q[0], q[1], q[2], q[3]);
            }
        } else {
            for (; i < numel; ++i) out[i] = (unsigned char)bn_ternary(gnk_load_scalar<DT>(w, i), dv, fast);
        }
    }
}

// int8 (any values) -> 2-bit fields: 16 values in, 4 bytes out per lane
__global__ void __launch_bounds__(BN_THREADS)
pack_i2_kernel(const signed char *__restrict__ x, unsigned char *__restrict__ out, int64_t numel) {
    const int64_t nchunks = numel / 16;
    const int64_t c = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x;
    if (c < nchunks) {
        const v4i raw = *(const v4i *)(x + c * 16);
        unsigned o = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned r = (unsigned)raw[j];
            o |= bn_pack4_i2((int)(r & 0xffu), (int)((r >> 8) & 0xffu), (int)((r >> 16) & 0xffu), (int)(r >> 24)) << (8 * j);
        }
        *(unsigned *)(out + c * 4) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nchunks * 16; i + 3 < numel; i += 4) out[i / 4] = (unsigned char)bn_pack4_i2(x[i], x[i + 1], x[i + 2], x[i + 3]);
}

__device__ __forceinline__ unsigned bn_unpack_byte(unsigned b) {   // one packed byte -> four sign-extended int8 (little-endian word)
    const int s = (int)(signed char)b;
    return bn_pack4_i8(s >> 6, (int)(signed char)(b << 2) >> 6, (int)(signed char)(b << 4) >> 6, (int)(signed char)(b << 6) >> 6);
}

// 4 packed bytes in, 16 int8 out per lane
__global__ void __launch_bounds__(BN_THREADS)
unpack_i2_kernel(const unsigned char *__restrict__ x, unsigned char *__restrict__ out, int64_t nbytes) {
    const int64_t nwords = nbytes / 4;
    const int64_t c = (int64_t)blockIdx.x * BN_THREADS + threadIdx.x;
    if (c < nwords) {
        const unsigned r = *(const unsigned *)(x + c * 4);
        *(v4i *)(out + c * 16) = (v4i){(int)bn_unpack_byte(r & 0xffu), (int)bn_unpack_byte((r >> 8) & 0xffu), (int)bn_unpack_byte((r >> 16) & 0xffu),
                                       (int)bn_unpack_byte(r >> 24)};
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nwords * 4; i < nbytes; ++i) *(unsigned *)(out + i * 4) = bn_unpack_byte(x[i]);
}

int partial_count(int64_t numel, int vec) {
    const int64_t trips = (numel / vec + BN_THREADS * BN_UNR - 1) / (BN_THREADS * BN_UNR);
    const int64_t cap = 4LL * gnk_num_cus() < BN_MAX_PARTIALS ? 4LL * gnk_num_cus() : BN_MAX_PARTIALS;
    return (int)(trips < 1 ? 1 : trips > cap ? cap : trips);
}

template <int DT> int launch_absmean(const void *x, int64_t numel, float *out, double *ws, hipStream_t st) {
    const int g = partial_count(numel, BnTraits<DT>::VEC);
    hipLaunchKernelGGL((absmean_partial_kernel<DT>), dim3(g), dim3(BN_THREADS), 0, st, (const char *)x, ws, numel);
    hipLaunchKernelGGL(absmean_final_kernel, dim3(1), dim3(BN_THREADS), 0, st, (const double *)ws, g, numel, out);
    return gnk_check_launch("bitnet_absmean");
}

template <int DT> int launch_quant(const void *w, const float *scale, float eps, void *out, int packed, int64_t numel, hipStream_t st) {
    constexpr int VEC = BnTraits<DT>::VEC;
    const int64_t blocks = (numel / VEC + BN_THREADS * BN_UNR - 1) / (BN_THREADS * BN_UNR);
    const dim3 grid((unsigned)(blocks < 1 ? 1 : blocks));
    if (packed) hipLaunchKernelGGL((bitnet_quant_kernel<DT, true>), grid, dim3(BN_THREADS), 0, st, (const char *)w, scale, eps, (unsigned char *)out, numel);
    else hipLaunchKernelGGL((bitnet_quant_kernel<DT, false>), grid, dim3(BN_THREADS), 0, st, (const char *)w, scale, eps, (unsigned char *)out, numel);
    return gnk_check_launch("bitnet_quantize_weight");
}

}  // namespace

extern "C" size_t gnk_bitnet_absmean_workspace_bytes(void) { return (size_t)BN_MAX_PARTIALS * sizeof(double); }

extern "C" int gnk_bitnet_absmean(const void *x, int x_dtype, int64_t numel, float *out, void *workspace, size_t workspace_bytes, void *stream) {
    if (!gnk_is_float_dtype(x_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_absmean: x must be bf16 / fp16 / fp32");
    if (numel <= 0) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_absmean: empty tensor (the mean of nothing)");
    if (!x || !out || !workspace) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_absmean: null pointer");
    if (workspace_bytes < gnk_bitnet_absmean_workspace_bytes()) return gnk_fail(GNK_ERR_WORKSPACE, "bitnet_absmean: workspace smaller than gnk_bitnet_absmean_workspace_bytes()");
    if (((uintptr_t)x & 15) || ((uintptr_t)workspace & 7) || ((uintptr_t)out & 3)) return gnk_fail(GNK_ERR_UNSUPPORTED, "bitnet_absmean: x must be 16-byte aligned");
    if (numel >= (1LL << 40)) return gnk_fail(GNK_ERR_UNSUPPORTED, "bitnet_absmean: tensor too large");
    int rc = gnk_init();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == GNK_BF16) return launch_absmean<GNK_BF16>(x, numel, out, (double *)workspace, st);
    if (x_dtype == GNK_F16) return launch_absmean<GNK_F16>(x, numel, out, (double *)workspace, st);
    return launch_absmean<GNK_F32>(x, numel, out, (double *)workspace, st);
}

extern "C" int gnk_bitnet_quantize_weight(const void *w, int w_dtype, int64_t numel, const float *scale, float eps, void *out, int packed, void *stream) {
    if (!gnk_is_float_dtype(w_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_quantize_weight: w must be bf16 / fp16 / fp32");
    if (numel < 0) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_quantize_weight: negative size");
    if (packed && numel % 4) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_quantize_weight: the packed form needs a multiple of 4 elements (got %lld)", (long long)numel);
    if (numel == 0) return GNK_OK;
    if (!w || !scale || !out) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_quantize_weight: null pointer");
    if (((uintptr_t)w & 15) || ((uintptr_t)out & (packed ? 1 : 7)) || ((uintptr_t)scale & 3))
        return gnk_fail(GNK_ERR_UNSUPPORTED, "bitnet_quantize_weight: w must be 16-byte aligned, the output 8-byte (packed: 2-byte) aligned");
    if (numel >= (1LL << 40)) return gnk_fail(GNK_ERR_UNSUPPORTED, "bitnet_quantize_weight: tensor too large");
    int rc = gnk_init();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (w_dtype == GNK_BF16) return launch_quant<GNK_BF16>(w, scale, eps, out, packed, numel, st);
    if (w_dtype == GNK_F16) return launch_quant<GNK_F16>(w, scale, eps, out, packed, numel, st);
    return launch_quant<GNK_F32>(w, scale, eps, out, packed, numel, st);
}

extern "C" int gnk_bitnet_pack_i2(const void *x, int64_t numel, void *out, void *stream) {
    if (numel < 0 || numel % 4) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_pack_i2: needs a multiple of 4 values (got %lld)", (long long)numel);
    if (numel == 0) return GNK_OK;
    if (!x || !out) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_pack_i2: null pointer");
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 3)) return gnk_fail(GNK_ERR_UNSUPPORTED, "bitnet_pack_i2: x must be 16-byte aligned, the output 4-byte aligned");
    int rc = gnk_init();
    if (rc) return rc;
    const int64_t blocks = (numel / 16 + BN_THREADS - 1) / BN_THREADS;
    hipLaunchKernelGGL(pack_i2_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(BN_THREADS), 0, (hipStream_t)stream, (const signed char *)x, (unsigned char *)out, numel);
    return gnk_check_launch("bitnet_pack_i2");
}

extern "C" int gnk_bitnet_unpack_i2(const void *x, int64_t nbytes, void *out, void *stream) {
    if (nbytes < 0) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_unpack_i2: negative size");
    if (nbytes == 0) return GNK_OK;
    if (!x || !out) return gnk_fail(GNK_ERR_BAD_ARG, "bitnet_unpack_i2: null pointer");
    if (((uintptr_t)x & 3) || ((uintptr_t)out & 15)) return gnk_fail(GNK_ERR_UNSUPPORTED, "bitnet_unpack_i2: x must be 4-byte aligned, the output 16-byte aligned");
    int rc = gnk_init();
    if (rc) return rc;
    const int64_t blocks = (nbytes / 4 + BN_THREADS - 1) / BN_THREADS;
    hipLaunchKernelGGL(unpack_i2_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(BN_THREADS), 0, (hipStream_t)stream, (const unsigned char *)x, (unsigned char *)out, nbytes);
    return gnk_check_launch("bitnet_unpack_i2");
}
