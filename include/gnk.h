/*
 * gnk.h -- C ABI of libgnk_hip.so: the MI355X (gfx950) kernels behind the gn_kernels operator
 * library and the quantized-training tensor subclasses.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers, int64 dims and strides in ELEMENTS, dtype codes from gnk_dtype;
 *   - inputs are borrowed and never written; outputs are caller-allocated, row-major contiguous
 *     (M, N) unless strides are given; nothing is allocated or freed, nothing synchronises;
 *   - `stream` is a hipStream_t (NULL = default stream); launches are asynchronous;
 *   - return value 0 on success, a negative gnk_status otherwise; gnk_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *
 * Citations are paths inside the reference repository (gau-nernst/quantized-training).
 */
#ifndef GNK_H
#define GNK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gnk_dtype {
    GNK_BF16 = 0,
    GNK_F16 = 1,
    GNK_F32 = 2,
    GNK_I8 = 3,
    GNK_E4M3 = 4, /* OCP float8_e4m3fn */
    GNK_E5M2 = 5, /* OCP float8_e5m2   */
    GNK_I32 = 6
} gnk_dtype;

typedef enum gnk_status {
    GNK_OK = 0,
    GNK_ERR_BAD_ARG = -1,     /* null pointer, negative size, unknown dtype code ...            */
    GNK_ERR_UNSUPPORTED = -2, /* legal request this build has no kernel for                     */
    GNK_ERR_LAUNCH = -3,      /* HIP reported an error at launch                                */
    GNK_ERR_WORKSPACE = -4    /* a required workspace pointer was NULL or too small             */
} gnk_status;

/* Library identification.  Rule: GNK_ABI_VERSION goes up by one whenever an entry point is added, removed, or changes its arguments or
 * the meaning of an argument or return code (kernel changes behind an unchanged signature do not count); a binding compares
 * gnk_abi_version() of the loaded library with the GNK_ABI_VERSION it was written against and refuses a mismatch.
 *   1: round 1.   2: rounds 2-3 -- gnk_scaled_mm_ws / gnk_scaled_mm_t, the fused quantize pair, gnk_quantize_int8_noalloc, the MX and BitNet
 *      entry points added; gnk_set_debug_flags removed from the shipped library; gnk_set_gemm_variant accepts 5, 6, -2 and -3; gnk_set_tile_rescale added. */
#define GNK_ABI_VERSION 2
int gnk_abi_version(void);

/* Message for the last non-zero status returned on this thread ("" if none). */
const char *gnk_last_error(void);

/* Optional one-time set-up for the current device (queries CU count).  Every entry point calls
 * it lazily; exposed so that a host can pay the cost outside its timed region. */
int gnk_init(void);

/*
 * Row-wise absmax symmetric INT8 quantization.
 * Replaces: subclasses/int8.py:12-26 (quantize_int8), which the reference runs as ~9 eager ATen
 * kernels or Inductor-generated Triton; here it is ONE pass (two for a strided reduction).
 *
 * x is a logical (R, C) matrix of x_dtype (BF16/F16/F32) with element strides (sxr, sxc); the
 * absmax runs over C for every r (the caller maps `dim` onto this by swapping strides).
 *   scale[r] = amax_c |x[r,c]| / 127            (fp32, true division), stored as x_dtype
 *   q[r,c]   = clip(round_half_even(x[r,c] / max(scale[r], eps)), -128, 127)
 * q has strides (sqr, sqc).  sr != 0 replaces the rounding by floor(v + u), u ~ U[0,1): 24 bits of word (e & 3) of the
 * Philox4x32-10 block with key `seed` and counter e >> 2, e = r*C + c + offset (the logical element number; four
 * elements share a block).  The reference draws torch.rand_like (int8.py:21): parity under SR is statistical.
 * workspace: needed only when the reduction runs along the strided dimension (sxc != 1); may be NULL otherwise.
 * gnk_quantize_int8_workspace_bytes() returns the size the library asks for: up to 32 rows of R fp32 words, the per-block
 * partial maxima of pass 1 -- no need to clear it, nothing is read that the call did not write.  The minimum accepted is
 * R * 4 bytes (one word per logical row); pass 1 then clears those words on `stream` and uses atomicMax.  Same bits either way.
 */
int gnk_quantize_int8(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int64_t sxc,
                      int8_t *q, int64_t sqr, int64_t sqc, void *scale, float eps,
                      int sr, uint64_t seed, uint64_t offset,
                      void *workspace, int64_t workspace_bytes, void *stream);

int64_t gnk_quantize_int8_workspace_bytes(int64_t R, int64_t C, int64_t sxr, int64_t sxc);

/*
 * Fused dual-direction quantization.  The reference quantizes every X, W and dY of a
 * mixed-precision Linear along BOTH dims each step (subclasses/mixed_precision.py:100-108 called from :154, :178,
 * :186) and notes the missed fusion itself (subclasses/int8.py:155-156); each call there is an independent
 * quantize_int8 (int8.py:12-26).  Same arithmetic, fewer HBM reads:
 *
 * gnk_quantize_int8_rowwise_colmax == gnk_quantize_int8 along the contiguous dim (rows of C contiguous elements,
 *   row stride sxr, 16-byte aligned) that ALSO writes colmax[c] = amax_r |x[r,c]| (fp32, C values) from the same read.
 *   Rows of at most 6144 16-bit / 3072 fp32 elements; workspace of ..._workspace_bytes() (0 = shape not supported:
 *   use gnk_quantize_int8), 16-byte aligned, holds one partial row of maxima per workgroup (no global atomics).
 * gnk_quantize_int8_colwise_given_absmax == gnk_quantize_int8 along the STRIDED dim of a row-major (P, Q) matrix
 *   (leading dimension ld; q with ldq) for a caller that already holds absmax[q] = amax_p |x[p,q]| (fp32): scale[q]
 *   = absmax[q] / 127 stored as x_dtype, q[p,q] = clip(round(x[p,q] / max(scale[q], eps))): one read instead of two.
 *   Stochastic rounding indexes Philox by q * P + p, as gnk_quantize_int8 does for the (Q, P) logical view.
 */
int64_t gnk_quantize_int8_rowwise_colmax_workspace_bytes(int x_dtype, int64_t R, int64_t C);
int gnk_quantize_int8_rowwise_colmax(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr,
                                     int8_t *q, int64_t sqr, void *scale, float *colmax, float eps,
                                     int sr, uint64_t seed, uint64_t offset,
                                     void *workspace, int64_t workspace_bytes, void *stream);
int gnk_quantize_int8_colwise_given_absmax(const void *x, int x_dtype, int64_t P, int64_t Q, int64_t ld,
                                           int8_t *q, int64_t ldq, void *scale, const float *absmax, float eps,
                                           int sr, uint64_t seed, uint64_t offset, void *stream);

/*
 * C(int32, M x N) = A(int8, M x K) @ B(int8, K x N), exact.
 * Replaces: kernels/gn_kernels/triton_mm.py:112-143 (int8_mm -> _triton_mm -> _matmul_kernel
 * :53-109 with ACC_DTYPE=int32).  Any strides; the three layouts of benchmark_mm.py:33-34,91-96
 * (A@B.T, A@B, A.T@B) take the MFMA fast path without materialising a transpose.
 */
int gnk_int8_mm(const int8_t *A, const int8_t *B, int32_t *C, int64_t M, int64_t N, int64_t K,
                int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, void *stream);

/*
 * Row/column-scaled low-precision GEMM with fused epilogue:
 *   order 0:  C[m,n] = cast_out( (f32(acc[m,n]) * row_scale[m]) * col_scale[n] )
 *   order 1:  C[m,n] = cast_out( (f32(acc[m,n]) * col_scale[n]) * row_scale[m] )
 * acc is int32 (exact) for I8 operands and fp32 for E4M3/E5M2 operands (the two fp8 formats may
 * be mixed).  col_scale is one scalar when col_scale_is_scalar != 0.  Scales are scale_dtype
 * (BF16/F16/F32), contiguous; C is out_dtype (BF16/F16/F32), row-major contiguous.
 * Replaces: triton_mm.py:382-407 (scaled_mm CUDA impl -> _scaled_mm_kernel :146-220), order 0;
 *           kernels/csrc/cutlass_sm89_fp8.cu:62-185 (scaled_fp8_mm), order 1;
 *           kernels/csrc/cutlass_sm120a_fp8.cu:106-217 (cutlass_scaled_fp8_mm), order 1.
 */
int gnk_scaled_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
                  const void *row_scale, const void *col_scale, int scale_dtype,
                  int col_scale_is_scalar, int order, int64_t M, int64_t N, int64_t K,
                  int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, void *stream);

/*
 * gnk_scaled_mm with an optional split-K workspace (no reference counterpart: Triton's autotuner has no split-K
 * config, triton_mm.py:16-50).  For int8 products with a long K and few output tiles -- the grad-weight GEMM
 * X.T @ dY of subclasses/mixed_precision.py:186 -- the K range is cut into chunks that run as separate workgroups;
 * int32 partial tiles go to `workspace` and a second kernel adds them (exactly) and applies the scale epilogue, so
 * the result is bit-identical to gnk_scaled_mm.  gnk_scaled_mm_workspace_bytes() returns the size the library
 * would like for a problem (0: it would not split); a NULL or smaller workspace simply runs gnk_scaled_mm.
 */
int64_t gnk_scaled_mm_workspace_bytes(int a_dtype, int b_dtype, int64_t M, int64_t N, int64_t K);
int gnk_scaled_mm_ws(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
                     const void *row_scale, const void *col_scale, int scale_dtype,
                     int col_scale_is_scalar, int order, int64_t M, int64_t N, int64_t K,
                     int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                     void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Tile-scaled (2-D block scale) GEMM: for ascending K-blocks kb of QK elements
 *   acc_f32[m,n] += (f32(partial_kb[m,n]) * sA[m / QM, kb]) * sB[kb, n / QN]
 * with a fresh int32 / fp32 partial accumulator per K-block; multiply, multiply, add are separate
 * roundings (no FMA).  sA is (M/QM, K/QK) with strides (ssam, ssak); sB is (K/QK, N/QN) with
 * strides (ssbk, ssbn); both scale_dtype.  C is out_dtype, row-major contiguous.
 * Replaces: triton_mm.py:410-441 (tile_scaled_mm CUDA impl -> _tile_scaled_mm_kernel :223-342).
 */
int gnk_tile_scaled_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
                       const void *sA, const void *sB, int scale_dtype,
                       int64_t M, int64_t N, int64_t K,
                       int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                       int64_t ssam, int64_t ssak, int64_t ssbk, int64_t ssbn,
                       int64_t QM, int64_t QN, int64_t QK, void *stream);

/*
 * Unscaled FP8 GEMM, fp32 accumulate, cast to out_dtype.
 * Replaces: kernels/csrc/cutlass_sm89_fp8.cu:28-57 (fp8_mm), cutlass_sm120a_fp8.cu:35-104
 * (cutlass_fp8_mm), and _triton_mm with fp8 inputs (triton_mm.py:131-143; benchmark_mm.py:129).
 */
int gnk_fp8_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
               int64_t M, int64_t N, int64_t K,
               int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, void *stream);

/*
 * OCP microscaling (MX) quantization, SURVEY.md section 8f rank 4.
 * Replaces: kernels/gn_kernels/utils.py:83-107 (quantize_mx; scale rules :30-41 "nv" and :45-49 "ocp"; FP4 rounding
 * table :52-80), eager ATen in the reference, ONE HBM-bound pass here.
 * x: (M, N) of x_dtype, row stride ldx elements, N % 32 == 0.  q_format: 7 = FP4 e2m1, two values per byte (even column
 * in the low nibble; q rows are N / 2 bytes), GNK_E4M3 or GNK_E5M2 (q rows are N bytes); ldq = row stride of q in BYTES.
 * scales: one E8M0 byte per row and 32 columns, row stride lds bytes.  method: 0 = "ocp" (floor), 1 = "nv" (ceil).
 */
int gnk_quantize_mx(const void *x, int x_dtype, int64_t M, int64_t N, int64_t ldx, void *q, int q_format,
                    int64_t ldq, void *scales, int64_t lds, int method, void *stream);

/*
 * MXFP4 GEMM:  C[m,n] = cast_out( sum_kb (A[m,kb] . B[n,kb]) * 2^(sA[m,kb]-127) * 2^(sB[n,kb]-127)  + bias[n] )
 * over blocks kb of 32 k.  A: (M, K / 2) bytes, row stride lda BYTES; B: the operand as B.T, (N, K / 2) bytes, row stride
 * ldb bytes (the reference requires B.T contiguous); scale_A (M, K / 32), scale_B (N, K / 32) E8M0 bytes with row strides
 * ldsa / ldsb; bias NULL or N values of bias_dtype, added in fp32 before the cast; C row-major contiguous out_dtype.
 * Replaces: kernels/gn_kernels/cutlass_mm.py:80-86 (mxfp4_mm) -> kernels/csrc/cutlass_sm120a_fp4.cu:42-136.  The scales
 * are taken in quantize_mx's plain layout, not in CUTLASS's Blackwell-specific shuffled order (pack_block_scales_nv).
 * K % 256 == 0 with 16-byte aligned rows runs on v_mfma_scale_f32_16x16x128_f8f6f4; anything else on a generic kernel.
 */
int gnk_mxfp4_mm(const void *A, const void *B, void *C, int out_dtype, const void *scale_A, const void *scale_B,
                 const void *bias, int bias_dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                 int64_t ldsa, int64_t ldsb, void *stream);

/*
 * MXFP8 GEMM:  C[m,n] = cast_out( sum_kb (A[m,kb] . B[n,kb]) * 2^(sA[m,kb]-127) * 2^(sB[n,kb]-127)  + bias[n] )  over blocks kb of 32 k.
 * A: (M, K) bytes of a_dtype (GNK_E4M3 / GNK_E5M2), row stride lda; B: the operand as B.T, (N, K) bytes of b_dtype, row stride ldb;
 * scale_A (M, K / 32), scale_B (N, K / 32) E8M0 bytes with row strides ldsa / ldsb, exactly what gnk_quantize_mx writes; bias / C as in
 * gnk_mxfp4_mm.  No counterpart in the reference: its quantize_mx (kernels/gn_kernels/utils.py:83-107) produces e4m3 / e5m2 MX operands,
 * but its only MX GEMMs are the FP4 ones (cutlass_mm.py:80-104); this is the consumer the format was missing.
 * K % 256 == 0 with 16-byte aligned rows runs on v_mfma_scale_f32_16x16x128_f8f6f4; anything else on a generic kernel.
 */
int gnk_mxfp8_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype, const void *scale_A,
                 const void *scale_B, const void *bias, int bias_dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                 int64_t ldsa, int64_t ldsb, void *stream);

/*
 * BitNet b1.58 weight path, SURVEY.md section 8f rank 4 (the COL_SCALE_SCALAR caller of scaled_mm and its FSDP2 hooks).
 * All tensors contiguous, walked flat; eager ATen chains in the reference, one HBM-bound kernel each here.
 *
 * gnk_bitnet_absmean: out[0] = mean(|f32(x)|) as fp32.  Replaces subclasses/bitnet.py:110-112 (get_bitnet_scale).  Fixed
 *   summation order (fp32 within a 16-byte chunk, fp64 above): deterministic, within 1e-6 relative of any other order.  workspace: at
 *   least gnk_bitnet_absmean_workspace_bytes() bytes, 8-byte aligned (per-workgroup partial sums; no need to clear it).
 * gnk_bitnet_quantize_weight: clip(round_half_even(f32(w) / max(scale[0], eps)), -1, 1).  Replaces subclasses/bitnet.py:115-118
 *   (quantize_bitnet_weight); scale is a DEVICE fp32 scalar (no host sync).  packed = 0: out is numel int8 values; packed = 1:
 *   out is numel / 4 bytes in the layout of gnk_bitnet_pack_i2, fusing subclasses/bitnet.py:88-89 (fsdp_pre_all_gather).
 * gnk_bitnet_pack_i2: four consecutive int8 values [a, b, c, d] -> one byte (a << 6) | ((b & 3) << 4) | ((c & 3) << 2) | (d & 3).
 *   Replaces subclasses/bitnet.py:185-195 (_pack_i2_in_i8; rows are multiples of 4, so the flat walk equals the row-wise one).
 * gnk_bitnet_unpack_i2: nbytes packed bytes -> 4 * nbytes sign-extended int8 values.  Replaces subclasses/bitnet.py:198-201.
 */
size_t gnk_bitnet_absmean_workspace_bytes(void);
int gnk_bitnet_absmean(const void *x, int x_dtype, int64_t numel, float *out, void *workspace, size_t workspace_bytes, void *stream);
int gnk_bitnet_quantize_weight(const void *w, int w_dtype, int64_t numel, const float *scale, float eps, void *out, int packed,
                               void *stream);
int gnk_bitnet_pack_i2(const void *x, int64_t numel, void *out, void *stream);
int gnk_bitnet_unpack_i2(const void *x, int64_t nbytes, void *out, void *stream);

/*
 * Test / tuning hook (no reference counterpart): force a GEMM kernel variant for subsequent
 * calls on this thread.  variant < 0 restores the built-in heuristic.  Returns the previous
 * value.  Variants are listed in csrc/gnk_gemm_dispatch.h.
 */
int gnk_set_gemm_variant(int variant);

/*
 * Rescale form of gnk_tile_scaled_mm's fast kernels for subsequent calls on this thread (returns the previous mode):
 *   0 (default): fp32 outputs use the reference interpreter's form acc += (f32(part) * sA) * sB, three separately rounded operations
 *      (triton_mm.py:327-331 run under TRITON_INTERPRET=1; goldens g4 are bit-exact), bf16 / fp16 outputs use
 *      acc = fma(f32(part), sA * sB, acc): within the reference's tolerance for this op (benchmark_mm.py:119-121: atol 1e-4, rtol 1e-2;
 *      fp8 :144-145: bf16 assert_close defaults), half the vector instructions, about 1.4x the throughput;
 *   1: the three-rounding form for every output dtype;   2: the fused form for every output dtype.
 * The generic fallback kernel (odd block sizes) always uses the three-rounding form.  No reference counterpart.
 */
int gnk_set_tile_rescale(int mode);

/*
 * gnk_quantize_int8 for callers that CANNOT allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference, whose quantize_int8 (subclasses/int8.py:12-26) is
 * eager ATen and allocates freely.  Same arguments and results as gnk_quantize_int8, bit for bit, minus the workspace:
 * the fp32 absmax words a strided reduction needs between its two kernels live in a GNK_NOALLOC_SCRATCH_BYTES buffer
 * inside the library's code object (one copy per GPU, held until the process ends).  The call performs no hipMalloc /
 * hipFree and calls no allocating routine.  Limits: an input that takes the strided-reduction kernels (sxr == 1, rows a multiple of the
 * vector width, aligned: the same test gnk_quantize_int8 applies) over more than gnk_quantize_int8_noalloc_max_rows() =
 * GNK_NOALLOC_SCRATCH_BYTES / 4 logical rows is rejected with GNK_ERR_UNSUPPORTED (inputs the generic kernel takes need no scratch and
 * are never rejected for their size).  Under HIP graph capture: call gnk_init() once per device BEFORE the capture begins -- the first
 * call on a device runs hipGetDeviceCount / hipDeviceGetAttribute / hipGetSymbolAddress, which are not capturable.  Calls that overlap on one GPU (different streams or threads) share the buffer and RACE --
 * serialise them, or use gnk_quantize_int8 with one workspace per stream.  Every word a call reads it has written
 * (partial maxima) or cleared on `stream` (reductions too large for the buffer to hold partial rows) first.
 */
#define GNK_NOALLOC_SCRATCH_BYTES (256 * 1024)
int64_t gnk_quantize_int8_noalloc_max_rows(void);
int gnk_quantize_int8_noalloc(const void *x, int x_dtype, int64_t R, int64_t C, int64_t sxr, int64_t sxc,
                              int8_t *q, int64_t sqr, int64_t sqc, void *scale, float eps,
                              int sr, uint64_t seed, uint64_t offset, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GNK_H */
