// gnk_common.h -- shared host/device helpers for libgnk_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/gnk.h"

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));

#define GNK_LDS __attribute__((address_space(3)))
#define GNK_GLOBAL __attribute__((address_space(1)))

// ---------------------------------------------------------------- host-side error plumbing
int gnk_fail(int code, const char *fmt, ...);   // records a thread-local message, returns code
int gnk_check_launch(const char *what);        // hipGetLastError -> GNK_ERR_LAUNCH
// Timing-ablation flags (some make results WRONG on purpose) exist only in builds made with -DGNK_ABLATE (`make ABLATE=1`,
// tools/ablate*.py); the released library has neither the setter nor the tests in its kernels.
#ifdef GNK_ABLATE
int gnk_debug_flags_get(void);                  // flags of the calling thread (gnk_set_debug_flags)
#define GNK_DEBUG_BITS(v) (v)
#else
static inline int gnk_debug_flags_get(void) { return 0; }
#define GNK_DEBUG_BITS(v) 0
#endif
int gnk_quantize_noalloc_setup(void);          // gnk_quantize.hip: per-device hipGetSymbolAddress of its scratch array (idempotent)
int gnk_forced_variant_get(void);          // gnk_api.hip: the calling thread's gnk_set_gemm_variant value (-1: heuristic)
int gnk_num_cus(void);                         // CU count of the current device (cached)

static inline bool gnk_is_float_dtype(int d) { return d == GNK_BF16 || d == GNK_F16 || d == GNK_F32; }
static inline bool gnk_is_f8(int d) { return d == GNK_E4M3 || d == GNK_E5M2; }
static inline int gnk_dtype_size(int d) {
    switch (d) {
    case GNK_BF16: case GNK_F16: return 2;
    case GNK_F32: case GNK_I32: return 4;
    default: return 1;
    }
}

// ---------------------------------------------------------------- device numeric helpers
// bf16 <-> f32 by bit manipulation: RNE, NaN stays NaN (matches torch's cast).
__device__ __forceinline__ float gnk_bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// fp32 -> bf16, round to nearest even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (one instruction per TWO values; the
// integer-arithmetic version cost 8 VALU instructions per value and made the GEMM epilogue VALU-bound).
typedef __bf16 gnk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gnk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gnk_pack2_bf16(float lo, float hi) {
    const gnk_bf16x2 r = __builtin_convertvector((gnk_f32x2){lo, hi}, gnk_bf16x2);
    unsigned u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}
__device__ __forceinline__ unsigned short gnk_f32_to_bf16(float f) {
    const __bf16 r = (__bf16)f;
    unsigned short u;
    __builtin_memcpy(&u, &r, 2);
    return u;
}

__device__ __forceinline__ float gnk_f16_to_f32(unsigned short h) {
    _Float16 x;
    __builtin_memcpy(&x, &h, 2);
    return (float)x;
}

__device__ __forceinline__ unsigned gnk_pack2_f16(float lo, float hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 r = __builtin_convertvector((gnk_f32x2){lo, hi}, h2);   // round to nearest even (two v_cvt_f16_f32 + pack)
    unsigned u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}

__device__ __forceinline__ unsigned short gnk_f32_to_f16(float f) {
    _Float16 x = (_Float16)f;   // v_cvt_f16_f32: RNE
    unsigned short h;
    __builtin_memcpy(&h, &x, 2);
    return h;
}

template <int DT> __device__ __forceinline__ float gnk_load_scalar(const void *p, int64_t i) {
    if constexpr (DT == GNK_BF16) return gnk_bf16_to_f32(((const unsigned short *)p)[i]);
    else if constexpr (DT == GNK_F16) return gnk_f16_to_f32(((const unsigned short *)p)[i]);
    else return ((const float *)p)[i];
}

// raw bits now, float later: lets a load stay in flight until its value is needed
template <int DT> __device__ __forceinline__ unsigned gnk_load_raw(const void *p, int64_t i) {
    if constexpr (DT == GNK_F32) return ((const unsigned *)p)[i];
    else return ((const unsigned short *)p)[i];
}
template <int DT> __device__ __forceinline__ float gnk_raw_to_f32(unsigned raw) {
    if constexpr (DT == GNK_BF16) return gnk_bf16_to_f32((unsigned short)raw);
    else if constexpr (DT == GNK_F16) return gnk_f16_to_f32((unsigned short)raw);
    else return __uint_as_float(raw);
}

// Branch-free variants for a RUN-TIME dtype (bf16 / fp16 / fp32): two 2-byte loads -- the upper half of an fp32 element,
// or the same 2 bytes again for the 16-bit types (never past the element) -- and a select at conversion time.  Used
// where a load and its conversion sit in different loop iterations: uniform branches there produced control flow that
// hipcc 7.2 miscompiled under amdgpu_waves_per_eu (the fp32 arm skipped half of the conversions).
__device__ __forceinline__ unsigned gnk_load_raw_rt(const void *p, int dt, int64_t i) {
    const int w = dt == GNK_F32 ? 2 : 1;
    const unsigned short *h = (const unsigned short *)p + i * w;
    return (unsigned)h[0] | ((unsigned)h[w - 1] << 16);
}
// e2m1 (FP4) code -> value: {0, .5, 1, 1.5, 2, 3, 4, 6} with the sign in bit 3 (kernels/gn_kernels/utils.py:114-118 of the reference)
__device__ __forceinline__ float gnk_fp4_to_f32(unsigned c) {
    const unsigned m = c & 7u;
    const float mag = m < 2u ? 0.5f * (float)m : __uint_as_float(((126u + (m >> 1)) << 23) | ((m & 1u) << 22));   // 2^(e-1) * (1 + mant / 2)
    return (c & 8u) ? -mag : mag;
}

__device__ __forceinline__ float gnk_raw_to_f32_rt(unsigned raw, int dt) {
    const float f32 = __uint_as_float(raw), bf = __uint_as_float(raw << 16), hf = gnk_f16_to_f32((unsigned short)(raw & 0xffffu));
    return dt == GNK_F32 ? f32 : dt == GNK_F16 ? hf : bf;
}

__device__ __forceinline__ float gnk_load_scalar_rt(const void *p, int dt, int64_t i) {
    if (dt == GNK_BF16) return gnk_bf16_to_f32(((const unsigned short *)p)[i]);
    if (dt == GNK_F16) return gnk_f16_to_f32(((const unsigned short *)p)[i]);
    return ((const float *)p)[i];
}

__device__ __forceinline__ void gnk_store_scalar_rt(void *p, int dt, int64_t i, float v) {
    if (dt == GNK_BF16) ((unsigned short *)p)[i] = gnk_f32_to_bf16(v);
    else if (dt == GNK_F16) ((unsigned short *)p)[i] = gnk_f32_to_f16(v);
    else ((float *)p)[i] = v;
}

// Philox4x32-10; the CPU oracle (oracle/gnk_oracle.c) carries the same generator so that the
// stochastic-rounding kernels can be checked bit-for-bit.
__device__ __forceinline__ void gnk_philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        unsigned n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Stochastic-rounding stream: logical element `idx` of a call takes word (e & 3) of the Philox4x32-10 block with counter e >> 2,
// e = idx + offset -- FOUR elements per block (one block per element made the SR quantize passes VALU-bound: ~70 integer
// instructions per element, +104 ms on a Llama-1B step).  U[0,1) with 24 random bits.
__device__ __forceinline__ float gnk_sr_bits_to_uniform(unsigned w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float gnk_sr_uniform(unsigned long long seed, unsigned long long offset,
                                                unsigned long long idx) {
    const unsigned long long e = idx + offset, ctr = e >> 2;
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
    gnk_philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned k = (unsigned)e & 3u;
    return gnk_sr_bits_to_uniform(k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3]);
}

// the four uniforms of elements idx0 .. idx0 + 3; the caller guarantees (idx0 + offset) % 4 == 0
__device__ __forceinline__ void gnk_sr_uniform4(unsigned long long seed, unsigned long long offset, unsigned long long idx0, float (&u)[4]) {
    const unsigned long long ctr = (idx0 + offset) >> 2;
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
    gnk_philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = gnk_sr_bits_to_uniform(c[k]);
}

// stochastic rounding of an already divided value: floor(v + u), clipped (subclasses/int8.py:21,25)
// NaN quotients (NaN / x, x / NaN, Inf / Inf) become 0: the reference casts NaN to int8 there, which is implementation-defined (0 on
// x86 and on NVIDIA's saturating conversion); without the select fmaxf would turn them into -128.
__device__ __forceinline__ int gnk_sr_clip_i8(float v, float u) {
    const int r = (int)fminf(fmaxf(floorf(v + u), -128.0f), 127.0f);
    return v == v ? r : 0;
}

// One quantization step of subclasses/int8.py:18-25 given the divisor max(scale, eps).
template <bool SR>
__device__ __forceinline__ int gnk_quant_one(float x, float div, unsigned long long seed,
                                             unsigned long long offset, unsigned long long idx) {
    float v = x / div;                     // true IEEE division (no fast-math in this build)
    if constexpr (SR) v = floorf(v + gnk_sr_uniform(seed, offset, idx));
    else v = rintf(v);                     // v_rndne_f32: round half to even
    const int r = (int)fminf(fmaxf(v, -128.0f), 127.0f);
    return v == v ? r : 0;                 // NaN quotient -> 0 (see gnk_sr_clip_i8)
}

// Division by a divisor that many elements share.  A true fp32 division is ~10 VALU instructions
// (v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup), half of which depend on the divisor only; at 3 B of
// HBM traffic per element that made the quantize kernels VALU-bound (21 VALU instructions per element).
// GnkDivisor keeps the refined reciprocal; quot() is the remaining per-element part of the SAME sequence the
// compiler emits for `x / d` (q0 = x*r; two residual-correction FMAs pairs), minus the power-of-two pre-scaling
// and the special-value fix-up, which cannot change the result while d is in [2^-60, 2^60] and x is finite with
// |x / d| < 2^24 (ok() says whether d qualifies; callers fall back to `/` otherwise -- inf / NaN / tiny scales).
struct GnkDivisor {
    float d, r;
    __device__ __forceinline__ void set(float div) {
        d = div;
        const float r0 = __builtin_amdgcn_rcpf(div);
        r = __builtin_fmaf(__builtin_fmaf(-div, r0, 1.0f), r0, r0);
    }
    __device__ __forceinline__ bool ok() const { return d >= 0x1p-60f && d <= 0x1p60f; }   // false for NaN
    __device__ __forceinline__ float quot(float x) const {
        const float q0 = x * r;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, x), r, q0);
        return __builtin_fmaf(__builtin_fmaf(-d, q1, x), r, q1);
    }
};

__device__ __forceinline__ int gnk_rne_clip_i8(float v) { return (int)fminf(fmaxf(rintf(v), -128.0f), 127.0f); }   // finite v (GnkDivisor fast path)
// the true-division path also sees NaN / Inf rows: NaN quotient -> 0 (see gnk_sr_clip_i8)
__device__ __forceinline__ int gnk_rne_clip_i8_any(float v) { return v == v ? gnk_rne_clip_i8(v) : 0; }
// max(scale, eps) as torch.clamp_min computes it: a NaN scale stays NaN (fmaxf alone would return eps and hide the NaN row)
__device__ __forceinline__ float gnk_clamp_min_nan(float s, float eps) { return s == s ? fmaxf(s, eps) : s; }

// Running |x| maximum that PROPAGATES NaN, as the reference's abs().amax() does (subclasses/int8.py:17): kept as bit patterns, which order
// like the magnitudes for every non-negative float, Inf and NaN included (NaN patterns exceed Inf's).  16-bit inputs: two packed patterns
// per word (v_and + v_pk_max_u16 per TWO elements, cheaper than unpack + v_max_f32); fp32: one pattern per word.
typedef unsigned short gnk_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gnk_pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(gnk_us2, a), __builtin_bit_cast(gnk_us2, b)));
}
template <int DT> struct GnkAbsMax {
    unsigned m = 0u;
    __device__ __forceinline__ void add_word(unsigned w) {      // one dword of the input: 2 x 16-bit elements or 1 x fp32
        if constexpr (DT == GNK_F32) { const unsigned a = w & 0x7fffffffu; m = a > m ? a : m; }
        else m = gnk_pk_max_u16(m, w & 0x7fff7fffu);
    }
    __device__ __forceinline__ void add16(const v4i &raw) {
#pragma unroll
        for (int j = 0; j < 4; ++j) add_word((unsigned)raw[j]);
    }
    // this lane's maximum as an fp32 bit pattern (order-preserving, so lanes / waves can go on reducing with integer max)
    __device__ __forceinline__ unsigned bits() const {
        if constexpr (DT == GNK_F32) return m;
        const unsigned lo = m & 0xffffu, hi = m >> 16, h = lo > hi ? lo : hi;
        if constexpr (DT == GNK_BF16) return h << 16;
        else return __float_as_uint(gnk_f16_to_f32((unsigned short)h));
    }
};

// 16-byte store of a finished output tile row segment (the LDS-staged epilogues of the GEMM kernels).  Default: non-temporal.  Measured at
// 4096^3 on the one-wave-per-SIMD kernel (tools/store_mode_ab.sh, profiles/r03k_*): plain write-back stores 55.8-57.3 us per launch,
// nt / sc1 / sc0 sc1 / sc1 nt all 53.5-54.0, no store at all 55.4 on the box where plain was 57.6: with write-back stores the launch ends on
// the write-back of the 32 MiB of dirty lines it left in the L2s; streamed out, that traffic overlaps the epilogue itself.
// nt is the one flavour with a compiler builtin: the inline-asm forms (sc bits) hide the store from the compiler's hazard recognizer -- a
// VALU write to the data registers in the two cycles behind a 16-byte store corrupts it (62 wrong outputs in the fp8 golden test) -- so
// they carry their own s_nop and stay A/B builds.
// GNK_STORE_MODE: 0 plain write-back; 1 non-temporal; 2 sc1; 3 sc0 sc1 (system scope); 4 sc1 nt; 5 no store (timing ablation, results wrong).
#ifndef GNK_STORE_MODE
#define GNK_STORE_MODE 1
#endif
__device__ __forceinline__ void gnk_store16_out(void *ptr, const v4i &d) {
#if GNK_STORE_MODE == 0
    *(v4i *)ptr = d;
#elif GNK_STORE_MODE == 1
    __builtin_nontemporal_store(d, (v4i *)ptr);
#elif GNK_STORE_MODE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(d) : "memory");
#elif GNK_STORE_MODE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(ptr), "v"(d) : "memory");
#elif GNK_STORE_MODE == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(ptr), "v"(d) : "memory");
#else   // 5: timing ablation -- no store at all (results WRONG): what the output write costs a launch, tail included
    asm volatile("" : : "v"(ptr), "v"(d));
#endif
}
