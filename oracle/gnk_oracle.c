/*
 * gnk_oracle.c -- CPU restatement of the reference's scaled low-precision matmul path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (quantized-training_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (integer / fp32 outputs)
 * against golden vectors produced by importing the reference itself in the build container
 * (tools/gen_golden.py writes tests/golden; checked by tests/test_oracle_golden.py).
 *
 * Each function cites the reference lines it restates (paths relative to the reference repo).
 * Plain C99, scalar, strides in ELEMENTS, no fused multiply-add (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* dtype codes (same numbering as include/gnk.h so the tests can pass one enum around) */
enum { ORC_BF16 = 0, ORC_F16 = 1, ORC_F32 = 2, ORC_I8 = 3, ORC_E4M3 = 4, ORC_E5M2 = 5, ORC_I32 = 6 };

/* ------------------------------------------------------------------ scalar format helpers */
static float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

/* round-to-nearest-even, NaN stays NaN (torch's float->bfloat16 cast) */
static uint16_t f32_to_bf16(float f) {
    uint32_t u = f32_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return bits_f32(sign);
        float v = ldexpf((float)m, -24);
        return sign ? -v : v;
    }
    if (e == 31) return bits_f32(sign | 0x7f800000u | (m << 13));
    return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

static uint16_t f32_to_f16(float f) {
    uint32_t u = f32_bits(f);
    uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* >= 65520 rounds to inf */
    if (a < 0x33000001u) return sign;                        /* <= 2^-25 rounds to 0 */
    if (a < 0x38800000u) {                                   /* subnormal half */
        float av = bits_f32(a);
        float r = nearbyintf(ldexpf(av, 24));
        return (uint16_t)(sign | (uint16_t)r);
    }
    uint32_t mant = a & 0x7fffffu, e = (a >> 23) - 112u;
    uint32_t h = (e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

/* OCP float8_e4m3fn: 1-4-3, bias 7, no inf, S.1111.111 = NaN */
static float e4m3_to_f32(uint8_t b) {
    int s = b >> 7, e = (b >> 3) & 0xf, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf((float)(8 + m), e - 10);
    return s ? -v : v;
}

/* OCP float8_e5m2: 1-5-2, bias 15, IEEE inf/NaN */
static float e5m2_to_f32(uint8_t b) {
    int s = b >> 7, e = (b >> 2) & 0x1f, m = b & 3;
    float v;
    if (e == 31) v = m ? NAN : INFINITY;
    else if (e == 0) v = ldexpf((float)m, -16);
    else v = ldexpf((float)(4 + m), e - 17);
    return s ? -v : v;
}

static float load_float(const void *p, int dtype, int64_t i) {
    switch (dtype) {
    case ORC_BF16: return bf16_to_f32(((const uint16_t *)p)[i]);
    case ORC_F16: return f16_to_f32(((const uint16_t *)p)[i]);
    default: return ((const float *)p)[i];
    }
}

static void store_float(void *p, int dtype, int64_t i, float v) {
    switch (dtype) {
    case ORC_BF16: ((uint16_t *)p)[i] = f32_to_bf16(v); break;
    case ORC_F16: ((uint16_t *)p)[i] = f32_to_f16(v); break;
    default: ((float *)p)[i] = v; break;
    }
}

static float load_operand(const void *p, int dtype, int64_t i) {
    switch (dtype) {
    case ORC_I8: return (float)((const int8_t *)p)[i];
    case ORC_E4M3: return e4m3_to_f32(((const uint8_t *)p)[i]);
    default: return e5m2_to_f32(((const uint8_t *)p)[i]);
    }
}

/* ------------------------------------------------------------------ stochastic-rounding noise
 * The reference draws torch.rand_like (subclasses/int8.py:21); that stream cannot be reproduced
 * on another backend, so parity with the reference is statistical.  The HIP kernel and this
 * oracle share ONE counter-based generator (Philox4x32-10, key = seed, counter = element index
 * in the logical (R, C) row-major order + offset) so that HIP-vs-oracle stays bit-exact. */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* U[0,1) with 24 random bits for logical element index idx: word (e & 3) of the Philox block with counter e >> 2,
 * e = idx + offset (four elements share one block; the HIP kernels draw them together). */
float orc_sr_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint64_t e = idx + offset, ctr = e >> 2;
    uint32_t c[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(c[e & 3] >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ quantize_int8
 * subclasses/int8.py:12-26.  x is a logical (R, C) matrix with element strides (sxr, sxc); the
 * reduction runs over C for each r (dim=1 of the view the caller built).  scale_out has R
 * entries of x's dtype.  sr_mode: 0 = round-half-even (:23), 1 = floor(x + u) with u from the
 * shared Philox stream (:21), 2 = floor(x + noise[r*C+c]) with caller-supplied noise (used to
 * replay the reference's torch.rand_like draw in the golden test). */
int orc_quantize_int8(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int64_t sxc,
                      int8_t *q, int64_t sqr, int64_t sqc, void *scale_out, float eps, int sr_mode,
                      uint64_t seed, uint64_t offset, const float *noise) {
    for (int64_t r = 0; r < R; ++r) {
        float amax = 0.0f;                                   /* :17  abs().amax(dim) */
        for (int64_t c = 0; c < C; ++c) {
            float v = fabsf(load_float(x, x_dtype, r * sxr + c * sxc));
            if (v > amax || v != v) amax = v;
        }
        float scale = amax / 127.0f;                         /* :17  / 127 */
        float div = scale < eps ? eps : scale;               /* :18  scale.clip(eps) */
        for (int64_t c = 0; c < C; ++c) {
            float v = load_float(x, x_dtype, r * sxr + c * sxc) / div;   /* :18 true division */
            if (sr_mode == 0) v = nearbyintf(v);             /* :23 round-half-even */
            else {
                float u = sr_mode == 1 ? orc_sr_uniform(seed, offset, (uint64_t)(r * C + c))
                                       : noise[r * C + c];
                v = floorf(v + u);                           /* :21 */
            }
            if (v < -128.0f) v = -128.0f;                    /* :25 clip(-128, 127) */
            if (v > 127.0f) v = 127.0f;
            /* NaN quotient (NaN row, Inf / Inf): the reference casts NaN to int8, implementation-defined; this build defines it as 0 */
            q[r * sqr + c * sqc] = v != v ? (int8_t)0 : (int8_t)v;
        }
        store_float(scale_out, x_dtype, r, scale);           /* :26 scale.to(dtype) */
    }
    return 0;
}

/* ------------------------------------------------------------------ int8_mm
 * kernels/gn_kernels/triton_mm.py:53-109,126-128 with ACC_DTYPE=int32: exact integer GEMM. */
int orc_int8_mm(const int8_t *A, const int8_t *B, int32_t *Cout, int64_t M, int64_t N, int64_t K,
                int64_t sam, int64_t sak, int64_t sbk, int64_t sbn) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            int32_t acc = 0;
            for (int64_t k = 0; k < K; ++k)
                acc += (int32_t)A[m * sam + k * sak] * (int32_t)B[k * sbk + n * sbn];
            Cout[m * N + n] = acc;
        }
    return 0;
}

/* dot product of one (m, n) over k in [k0, k1): int32 exact for int8, fp32 sequential for fp8.
 * For fp8 the reference's accumulation order is hardware-defined (tl.dot), so fp8 parity is
 * tolerance-based (benchmark_mm.py:128-145). */
static float dot_as_f32(const void *A, int a_dtype, const void *B, int b_dtype, int64_t m, int64_t n,
                        int64_t k0, int64_t k1, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn) {
    if (a_dtype == ORC_I8) {
        const int8_t *a = (const int8_t *)A, *b = (const int8_t *)B;
        int32_t acc = 0;
        for (int64_t k = k0; k < k1; ++k) acc += (int32_t)a[m * sam + k * sak] * (int32_t)b[k * sbk + n * sbn];
        return (float)acc;                                   /* int32 -> f32, RNE (triton_mm.py:216) */
    }
    float acc = 0.0f;
    for (int64_t k = k0; k < k1; ++k)
        acc += load_operand(A, a_dtype, m * sam + k * sak) * load_operand(B, b_dtype, k * sbk + n * sbn);
    return acc;
}

/* ------------------------------------------------------------------ scaled_mm (row x col scale)
 * triton_mm.py:146-220,382-407: C[m,n] = cast((f32(acc) * row_scale[m]) * col_scale[n]);
 * col_scale may be a single scalar (COL_SCALE_SCALAR, :211-213).  order=1 gives the CUTLASS
 * epilogue order (acc * col) * row of kernels/csrc/cutlass_sm89_fp8.cu:100-107. */
int orc_scaled_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *Cout, int out_dtype,
                  const void *row_scale, const void *col_scale, int scale_dtype, int col_scalar,
                  int order, int64_t M, int64_t N, int64_t K,
                  int64_t sam, int64_t sak, int64_t sbk, int64_t sbn) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float rs = load_float(row_scale, scale_dtype, m);
        for (int64_t n = 0; n < N; ++n) {
            float cs = load_float(col_scale, scale_dtype, col_scalar ? 0 : n);
            float acc = dot_as_f32(A, a_dtype, B, b_dtype, m, n, 0, K, sam, sak, sbk, sbn);
            float v = order == 0 ? (acc * rs) * cs : (acc * cs) * rs;
            store_float(Cout, out_dtype, m * N + n, v);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ tile_scaled_mm
 * triton_mm.py:223-342,410-441: for ascending K-blocks kb of QK:
 *   acc_f32 += (f32(mma_kb) * sA[m/QM, kb]) * sB[kb, n/QN]        (:314-331)
 * separate multiply, multiply, add (no FMA): matches the reference kernel run under the Triton
 * interpreter, which is what the golden vectors hold. */
int orc_tile_scaled_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *Cout, int out_dtype,
                       const void *sA, const void *sB, int scale_dtype,
                       int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                       int64_t ssam, int64_t ssak, int64_t ssbk, int64_t ssbn,
                       int64_t QM, int64_t QN, int64_t QK) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int64_t kb = 0; kb * QK < K; ++kb) {
                int64_t k1 = (kb + 1) * QK < K ? (kb + 1) * QK : K;
                float part = dot_as_f32(A, a_dtype, B, b_dtype, m, n, kb * QK, k1, sam, sak, sbk, sbn);
                float a = load_float(sA, scale_dtype, (m / QM) * ssam + kb * ssak);
                float b = load_float(sB, scale_dtype, kb * ssbk + (n / QN) * ssbn);
                float t = part * a;
                t = t * b;
                acc = acc + t;
            }
            store_float(Cout, out_dtype, m * N + n, acc);
        }
    return 0;
}

/* plain fp8 GEMM -> out dtype (kernels/gn_kernels/cutlass_mm.py:36-47, csrc/cutlass_sm89_fp8.cu:28-57;
 * also _triton_mm with fp8 inputs, benchmark_mm.py:128-129) */
int orc_fp8_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *Cout, int out_dtype,
               int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n)
            store_float(Cout, out_dtype, m * N + n,
                        dot_as_f32(A, a_dtype, B, b_dtype, m, n, 0, K, sam, sak, sbk, sbn));
    return 0;
}

/* format-conversion probes so the tests can pin the helpers against torch's casts */
float orc_bf16_to_f32(uint16_t h) { return bf16_to_f32(h); }
uint16_t orc_f32_to_bf16(float f) { return f32_to_bf16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_e4m3_to_f32(uint8_t b) { return e4m3_to_f32(b); }
float orc_e5m2_to_f32(uint8_t b) { return e5m2_to_f32(b); }
