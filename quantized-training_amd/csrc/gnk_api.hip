// gnk_api.hip -- extern "C" entry points of libgnk_hip.so (see include/gnk.h) and the host-side
// dispatch between the LDS-DMA MFMA kernels and the generic fallback.
#include "gnk_gemm_dispatch.h"

#include <math.h>
#include <string.h>

#include <atomic>

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
static thread_local int g_forced_variant = -1;
static thread_local int g_tile_rescale = 0;   // gnk_set_tile_rescale: 0 auto, 1 exact, 2 fused
#ifdef GNK_ABLATE
static thread_local int g_debug_flags = 0;
#endif

int gnk_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int gnk_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gnk_fail(GNK_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return GNK_OK;
}

static std::atomic<int> g_cus[64];   // forward and autograd threads both come through here

#ifdef GNK_ABLATE
int gnk_debug_flags_get(void) { return g_debug_flags; }
#endif

int gnk_num_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = g_cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

extern "C" int gnk_abi_version(void) { return GNK_ABI_VERSION; }
extern "C" const char *gnk_last_error(void) { return g_err; }

extern "C" int gnk_init(void) {
    // every entry point calls this: after the first call on a device it is one hipGetDevice (thread-local state) and one flag load
    static std::atomic<bool> ready[64];
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && ready[dev].load(std::memory_order_acquire)) return GNK_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return gnk_fail(GNK_ERR_LAUNCH, "no HIP device is visible: libgnk_hip.so has no CPU path");
    gnk_num_cus();
    const int rc = gnk_quantize_noalloc_setup();   // device address of the built-in scratch buffer, once per device
    if (rc == GNK_OK && dev >= 0 && dev < 64) ready[dev].store(true, std::memory_order_release);
    return rc;
}

int gnk_forced_variant_get(void) { return g_forced_variant; }

extern "C" int gnk_set_gemm_variant(int variant) {
    const int old = g_forced_variant;
    g_forced_variant = variant;
    return old;
}

extern "C" int gnk_set_tile_rescale(int mode) {
    const int old = g_tile_rescale;
    g_tile_rescale = mode < 0 || mode > 2 ? 0 : mode;
    return old;
}

#ifdef GNK_ABLATE
// Timing-ablation hook of ABLATE builds only (results are WRONG while bits 0-3 are set; bits listed in tools/ablate.py).
extern "C" int gnk_set_debug_flags(int flags) {
    const int old = g_debug_flags;
    g_debug_flags = flags;
    return old;
}
#endif

// ------------------------------------------------------------------ GEMM dispatch
namespace {

struct OperandLayout {
    bool fast;     // usable by the LDS-DMA kernels
    bool kmajor;   // K contiguous (else M/N contiguous)
    int64_t ld;
};

// ptr: operand base; ext: M or N; s_mn / s_k: element strides along M(N) and K.
OperandLayout classify(const void *ptr, int64_t ext, int64_t K, int64_t s_mn, int64_t s_k) {
    OperandLayout o = {false, true, 0};
    if (((uintptr_t)ptr & 15) != 0) return o;
    // K-major: every row is a contiguous K-vector starting at a 16-byte boundary.  A single row (ext == 1) has no pitch to round K up to:
    // the K-tail stager fetches whole 16-byte chunks, so K itself must be a multiple of 16 there (else the generic kernel takes the call)
    if (s_k == 1 && (ext == 1 ? K % 16 == 0 : (s_mn % 16 == 0 && s_mn >= K))) {
        o.fast = true; o.kmajor = true; o.ld = ext == 1 ? K : s_mn;
        return o;
    }
    // M/N-major: every k is a contiguous M(N)-vector of a multiple of 16 elements
    if (s_mn == 1 && ext >= 16 && ext % 16 == 0 && (K == 1 || (s_k % 16 == 0 && s_k >= ext))) {
        o.fast = true; o.kmajor = false; o.ld = K == 1 ? ext : s_k;
        return o;
    }
    return o;
}

gnk_gemm_launcher pick_launcher(int op, int fa, int fb, int variant) {
    if (op == OP_I8)
        return variant == GNK_VARIANT_256 ? gnk_launch_i8_v0
               : variant == GNK_VARIANT_256_SIMPLE ? gnk_launch_i8_v3
               : variant == GNK_VARIANT_256_PP8 ? gnk_launch_i8_v4
               : variant == GNK_VARIANT_256_W4 ? gnk_launch_i8_v5
               : variant == GNK_VARIANT_64X128 ? gnk_launch_i8_v6
                                                : gnk_launch_i8_v1;
    if (variant == GNK_VARIANT_256_W4) {
        static const gnk_gemm_launcher f8w4[2][2] = {{gnk_launch_f8_00_v5, gnk_launch_f8_01_v5}, {gnk_launch_f8_10_v5, gnk_launch_f8_11_v5}};
        return f8w4[fa][fb];
    }
    if (variant == GNK_VARIANT_256_PP8) {
        static const gnk_gemm_launcher f8pp8[2][2] = {{gnk_launch_f8_00_v4, gnk_launch_f8_01_v4}, {gnk_launch_f8_10_v4, gnk_launch_f8_11_v4}};
        return f8pp8[fa][fb];
    }
    static const gnk_gemm_launcher f8[2][2][2] = {
        {{gnk_launch_f8_00_v0, gnk_launch_f8_00_v1}, {gnk_launch_f8_01_v0, gnk_launch_f8_01_v1}},
        {{gnk_launch_f8_10_v0, gnk_launch_f8_10_v1}, {gnk_launch_f8_11_v0, gnk_launch_f8_11_v1}}};
    return f8[fa][fb][variant == GNK_VARIANT_256 ? 0 : 1];
}

struct GemmCall {
    const void *A; int a_dtype; const void *B; int b_dtype; void *C; int out_dtype;
    const void *s0; const void *s1; int scale_dtype; int col_scalar; int order; int epi;
    int64_t M, N, K, sam, sak, sbk, sbn;
    int64_t ssam, ssak, ssbk, ssbn, QM, QN, QK;
    void *ws; int64_t ws_bytes;   // optional split-K workspace (gnk_scaled_mm_ws)
};

// ---- kernel choice: a cost model fitted to MI355X measurements (tools/cmp_variants.py, tools/splitk_probe.py) ----------
// Times in us.  A 256 x 256 workgroup (one per CU) costs ~15 fixed (launch, 4-stage prologue, 128 KiB result write) and
// ~1.4 per 128 k; a round of 128 x 128 workgroups (two per CU) ~6 fixed and ~1.0 per 128 k.  Split-K (int8 row/col
// epilogue, workspace offered): long-K products with few output tiles -- the grad-weight GEMM X.T @ dY of a Linear,
// M = in, N = out, K = tokens -- leave most CUs idle (1024 x 1024 x 8192: 64 tiles, 63 us); their K range is cut into
// `splits` chunks, each workgroup writes an int32 partial tile into its own slab of the workspace, and a second kernel
// adds the slabs (exactly: integers) and applies the scale epilogue (~6 us + 8 B per element and split at ~4 TB/s).
struct GemmPlan { int variant; int splits; int first_steps; };   // first_steps > 0: uneven two-way split, K-steps of split 0

// w4: the 256 x 256 launch would go to the one-wave-per-SIMD kernel (round 3: ~12 fixed, ~1.08 per 128 k -- 2048 x 5632 x 32768 unsplit in
// 288 us); a split must then promise 8 % to be taken (the uneven two-way split of that shape measured 292 us against 286-290 unsplit).
GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, bool can256, bool can128, bool can_split, bool w4) {
    const double fix256 = w4 ? 12.0 : 15.0, per256 = w4 ? 1.08 : 1.4, margin = w4 ? 1.08 : 1.0;
    const double cus = (double)gnk_num_cus();
    const double ksteps = (double)K / 128.0;
    GemmPlan best = {GNK_VARIANT_GENERIC, 1, 0};
    double best_t = 1e30;
    if (can256) {
        const double tiles = (double)(((M + 255) / 256) * ((N + 255) / 256));
        const double t = ceil(tiles / cus) * (fix256 + per256 * ksteps);
        if (t < best_t) { best_t = t; best = {GNK_VARIANT_256, 1, 0}; }
        // round 2: split-K on the four-stage 256 x 256 kernel too (grad-weights of square Linears, 2048 x 2048 x 32768: 64 tiles)
        if (can_split && K >= 8192) {
            for (int sp = 2; sp <= 8; sp *= 2) {
                if (K / sp < 2048) break;
                const double ts = ceil(tiles * sp / cus) * (fix256 + per256 * ksteps / sp) + 6.0 + (double)sp * (double)M * (double)N * 8.0 / 4.0e6;
                if (ts * margin < best_t) { best_t = ts * margin; best = {GNK_VARIANT_256, sp, 0}; }
            }
            // Uneven two-way split for tile counts between half the chip and the whole chip (grad-weights of the Llama-1B MLP,
            // 2048 x 5632 x 32768: 176 tiles on 256 CUs, where one tile's K loop is the floor of any unsplit launch).  Split 0 of every
            // tile is dispatched first and holds its CU to the end; the `idle` other CUs take the short second splits in
            // r = ceil(tiles / idle) rounds; s, the length of a short split, makes both sides end together:
            //     r * (15 + 1.4 s) = 15 + 1.4 (ksteps - s).
            if (tiles < cus && tiles > cus / 2) {
                const double idle = cus - tiles, r = ceil(tiles / idle);
                const double sshort = floor((fix256 + per256 * ksteps - fix256 * r) / (per256 * (r + 1.0)));
                if (sshort >= 16.0 && ksteps - sshort >= 16.0) {
                    const double ts = fix256 + per256 * (ksteps - sshort) + 6.0 + 2.0 * (double)M * (double)N * 8.0 / 4.0e6;
                    if (ts * margin < best_t) { best_t = ts * margin; best = {GNK_VARIANT_256, 2, (int)(ksteps - sshort)}; }
                }
            }
        }
    }
    if (can128) {
        const double tiles = (double)(((M + 127) / 128) * ((N + 127) / 128));
        const double t = ceil(tiles / (2.0 * cus)) * (6.0 + 1.0 * ksteps);
        if (t < best_t) { best_t = t; best = {GNK_VARIANT_128, 1, 0}; }
        if (can_split && K >= 4096) {
            for (int sp = 2; sp <= 8; sp *= 2) {
                if (K / sp < 512) break;
                const double ts = ceil(tiles * sp / (2.0 * cus)) * (6.0 + 1.0 * ksteps / sp) + 6.0 + (double)sp * (double)M * (double)N * 8.0 / 4.0e6;
                if (ts < best_t) { best_t = ts; best = {GNK_VARIANT_128, sp, 0}; }
            }
        }
    }
    return best;
}

template <int OUT>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const int *__restrict__ ws, void *__restrict__ C, const void *__restrict__ s0,
                                                            const void *__restrict__ s1, int64_t M, int64_t N, int splits, int scale_dtype,
                                                            int col_scalar, int order) {
    const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 4 consecutive n of one m (N % 4 == 0)
    const int64_t idx = quad * 4;
    if (idx >= M * N) return;
    const int64_t m = idx / N, n = idx - m * N;
    v4i acc = *(const v4i *)(ws + idx);
    for (int s = 1; s < splits; ++s) {
        const v4i part = *(const v4i *)(ws + (int64_t)s * M * N + idx);
        acc += part;
    }
    const float rs = gnk_load_scalar_rt(s0, scale_dtype, m);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float cs = gnk_load_scalar_rt(s1, scale_dtype, col_scalar ? 0 : n + r);
        const float a = (float)acc[r];
        v[r] = order == 0 ? __fmul_rn(__fmul_rn(a, rs), cs) : __fmul_rn(__fmul_rn(a, cs), rs);
    }
    if constexpr (OUT == GNK_F32) {
        *(v4f *)((float *)C + idx) = (v4f){v[0], v[1], v[2], v[3]};
    } else {
        v2i o;
        if constexpr (OUT == GNK_BF16) {
            o[0] = (int)gnk_pack2_bf16(v[0], v[1]);
            o[1] = (int)gnk_pack2_bf16(v[2], v[3]);
        } else {
            o[0] = (int)gnk_pack2_f16(v[0], v[1]);
            o[1] = (int)gnk_pack2_f16(v[2], v[3]);
        }
        *(v2i *)((unsigned short *)C + idx) = o;
    }
}

int run_gemm(const char *name, const GemmCall &c, hipStream_t st) {
    if (c.M < 0 || c.N < 0 || c.K < 0) return gnk_fail(GNK_ERR_BAD_ARG, "%s: negative dimension", name);
    const bool i8 = c.a_dtype == GNK_I8 && c.b_dtype == GNK_I8;
    const bool f8 = gnk_is_f8(c.a_dtype) && gnk_is_f8(c.b_dtype);
    if (!i8 && !f8) return gnk_fail(GNK_ERR_BAD_ARG, "%s: operands must be both int8 or both fp8 (got dtype codes %d, %d)", name, c.a_dtype, c.b_dtype);
    if (c.M == 0 || c.N == 0) return GNK_OK;
    if (!c.C || (c.K > 0 && (!c.A || !c.B))) return gnk_fail(GNK_ERR_BAD_ARG, "%s: null pointer", name);
    if (c.epi != EPI_PLAIN) {
        if (!c.s0 || !c.s1) return gnk_fail(GNK_ERR_BAD_ARG, "%s: null scale pointer", name);
        if (!gnk_is_float_dtype(c.scale_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "%s: scale dtype code %d is not bf16/f16/f32", name, c.scale_dtype);
    }
    if (!(c.epi == EPI_PLAIN && i8) && !gnk_is_float_dtype(c.out_dtype))
        return gnk_fail(GNK_ERR_BAD_ARG, "%s: output dtype code %d is not bf16/f16/f32", name, c.out_dtype);
    if (c.epi == EPI_TILE && (c.QM <= 0 || c.QN <= 0 || c.QK <= 0)) return gnk_fail(GNK_ERR_BAD_ARG, "%s: non-positive scale block size", name);
    int rc = gnk_init();
    if (rc) return rc;

    const int op = i8 ? OP_I8 : OP_F8;
    const int fa = c.a_dtype == GNK_E5M2 ? F8_E5M2 : F8_E4M3, fb = c.b_dtype == GNK_E5M2 ? F8_E5M2 : F8_E4M3;

    int variant = g_forced_variant;
    const OperandLayout la = classify(c.A, c.M, c.K, c.sam, c.sak);
    const OperandLayout lb = classify(c.B, c.N, c.K, c.sbn, c.sbk);
    const bool fast_layout = la.fast && lb.fast && c.M < (1LL << 31) && c.N < (1LL << 31) && la.ld * 256 < (1LL << 31) &&
                             lb.ld * 256 < (1LL << 31) && ((c.M + 127) / 128) * ((c.N + 127) / 128) < (1LL << 31);
    // 4-stage 256^2 kernel: 64-k stages, no tile-scaled epilogue; the two-buffer kernels step 128 k
    const bool ok_pp2 = fast_layout && c.K >= 64 && c.K % 64 == 0 && c.epi != EPI_TILE;
    // tile-scaled fast path: scale blocks of 64 or 32 k (2 or 4 per 128-k step) or of a multiple of 128 k, and scale offsets that fit 32-bit indices;
    // QN % 4 != 0 (per-column scales) on the 128-k instances only (GemmCfg::PERCOL)
    bool tile_ok = true;
    if (c.epi == EPI_TILE) {
        const int64_t amax = ((c.M - 1) / c.QM) * c.ssam + (c.K / c.QK) * c.ssak, bmax = (c.K / c.QK) * c.ssbk + ((c.N - 1) / c.QN) * c.ssbn;
        tile_ok = (c.QK % GNK_BK == 0 || c.QK == 64 || c.QK == 32) && c.QK / GNK_BK < (1 << 20) && c.K % c.QK == 0 && (c.QN % 4 == 0 || c.QK == GNK_BK) && c.N % 4 == 0 && c.ssam >= 0 && c.ssak >= 0 && c.ssbk >= 0 && c.ssbn >= 0 && amax < (1LL << 31) && bmax < (1LL << 31);
    }
    // the 128 x 128 kernel masks a K tail itself (any K; plain and row/col epilogues); the tile-scaled epilogue needs whole 128-k steps
    const bool ok_128 = fast_layout && c.K >= 1 && (c.K % GNK_BK == 0 || c.epi != EPI_TILE) && tile_ok;
    const bool whole_steps = c.K >= GNK_BK && c.K % GNK_BK == 0;
    if ((variant == GNK_VARIANT_256_SIMPLE && op != OP_I8) || ((variant == GNK_VARIANT_256_SIMPLE || variant == GNK_VARIANT_256_PP8 || variant == GNK_VARIANT_256_W4) && c.epi == EPI_TILE)) variant = GNK_VARIANT_256;
    const bool ok_w4 = ok_pp2 && c.K % 256 == 0;   // one-wave-per-SIMD kernel: four 64-k stages per loop iteration
    if (variant == GNK_VARIANT_256_W4 && !ok_w4) variant = GNK_VARIANT_256;
    // 64 x 128 tile: plain / row-col epilogue (any K: it masks a K tail like the 128 x 128 kernel)
    const bool ok_64 = ok_128 && c.epi != EPI_TILE;
    if (variant == GNK_VARIANT_64X128 && !ok_64) variant = GNK_VARIANT_128;
    if (variant == GNK_VARIANT_256 && c.epi == EPI_TILE) variant = GNK_VARIANT_128;
    int splits = 1, first_steps = 0;
    if (variant < 0 || variant > GNK_VARIANT_LAST) {
        const bool can_split = c.ws && op == OP_I8 && c.epi == EPI_ROWCOL && ok_128 && whole_steps && c.N % 4 == 0 && ((uintptr_t)c.C & 15) == 0 && ((uintptr_t)c.ws & 15) == 0;
        const GemmPlan plan = plan_gemm(c.M, c.N, c.K, ok_pp2, ok_128, can_split, ok_w4 && variant != GNK_VARIANT_AUTO_NO_W4);
        const bool allow_w4 = variant != GNK_VARIANT_AUTO_NO_W4;
        variant = plan.variant;
        if (plan.splits > 1 && c.ws_bytes >= (int64_t)plan.splits * c.M * c.N * 4) { splits = plan.splits; first_steps = plan.first_steps; }
        // Round 3: the one-wave-per-SIMD kernel takes every 256 x 256 launch it can (K a multiple of 256; split-K chunks likewise).  Measured at
        // 4096^3 against the 8-phase kernel on one box (tools/cmp_variants.py): int8 A@B.T 57.4 vs 57.9 us, A@B 56.6 vs 59.8, A.T@B 56.8 vs 64.0;
        // fp8 56.0 / 55.6 / 55.7 vs 55.8 / 57.6 / 61.2; at K = 8192 101.0-102.6 vs 103.7-117.5.  The forward GEMMs of training (A@B.T, more tiles
        // than CUs: 32768 x {2048, 5632} x 2048, x 2048 x 5632) are level with the 8-phase kernel since the output stores became non-temporal
        // (118-125 / 315-316 / 256 vs 120-124 / 325-329 / 261 us, profiles/r03p_*), so there is no exception for them.
        if (variant == GNK_VARIANT_256 && allow_w4 && ok_w4) {
            if (splits == 1) {
                variant = GNK_VARIANT_256_W4;
            } else if (first_steps > 0) {
                first_steps += first_steps & 1;
                if ((int64_t)first_steps * GNK_BK < c.K) variant = GNK_VARIANT_256_W4;
            } else if ((((c.K / GNK_BK + splits - 1) / splits) * GNK_BK) % 256 == 0) {
                variant = GNK_VARIANT_256_W4;
            }
        }
        // Without it (K % 256 != 0), A K-major (A@B.T, A@B): the 8-phase kernel.  Measured at 4096^3 against the 4-stage one (tools/cmp_variants.py,
        // round 2): int8 A@B.T 54.0 vs 59.4 us, A@B 56.2 vs 57.5; fp8 (one 16x16x128 MFMA per tile and step instead of two 32x32x64) A@B.T 52.2 vs
        // 59.4, A@B 54.1 vs 55.7.  A M-major (A.T@B) stays on the 4-stage kernel (fp8 55.3 vs 57.9, int8 level).
        // Exception (tools/pp_choice_probe.py): A@B with a narrow output and a long K (grad-input of the Llama gate / up projections,
        // 32768 x 2048 x 5632: 315 vs 337 us; x 8192: 446 vs 468) stays on the 4-stage kernel; at K = 2048 the 8-phase kernel wins there too.
        const bool narrow_long_nn = !lb.kmajor && c.N <= 2048 && c.K >= 4096 && c.M >= 4 * c.N;
        if (variant == GNK_VARIANT_256 && splits == 1 && la.kmajor && ok_128 && whole_steps && !narrow_long_nn) variant = GNK_VARIANT_256_PP8;
        // Round 3: few 128 x 128 tiles (at most one workgroup per CU, where a 128-k step takes 1.0 us instead of the 0.55 us two co-resident workgroups
        // reach): 64 x 128 tiles double the workgroups.  Measured (tools/mid_ab.py, one box): 1024^3 A@B.T 9.0 -> 7.5 us, A.T@B 10.1 -> 8.7;
        // 2048^3 A@B.T 15.2 -> 13.9 (A@B 16.9 -> 16.4, A.T@B 16.5 -> 17.2: not taken); from 512 tiles up the 128 x 128 tile wins (8192 x 1024 x 1024:
        // 11.6 vs 14.5 us).
        if (variant == GNK_VARIANT_128 && splits == 1 && ok_64 && g_forced_variant != GNK_VARIANT_AUTO_NO_64) {
            const int64_t tiles128 = ((c.M + 127) / 128) * ((c.N + 127) / 128), cus = gnk_num_cus();
            if (tiles128 * 2 <= cus || (tiles128 <= cus && la.kmajor && lb.kmajor)) variant = GNK_VARIANT_64X128;
        }
    } else if (variant != GNK_VARIANT_GENERIC &&
               !((variant == GNK_VARIANT_256 || variant == GNK_VARIANT_256_W4) ? ok_pp2 : (variant == GNK_VARIANT_128 || variant == GNK_VARIANT_64X128) ? ok_128 : (ok_128 && whole_steps))) {
        return gnk_fail(GNK_ERR_UNSUPPORTED, "%s: forced fast variant %d cannot take these strides / sizes", name, variant);
    }

    if (variant == GNK_VARIANT_GENERIC) {
        GenericParams g;
        memset(&g, 0, sizeof(g));
        g.A = (const char *)c.A; g.B = (const char *)c.B; g.C = c.C; g.s0 = c.s0; g.s1 = c.s1;
        g.M = c.M; g.N = c.N; g.K = c.K; g.sam = c.sam; g.sak = c.sak; g.sbk = c.sbk; g.sbn = c.sbn;
        g.ssam = c.ssam; g.ssak = c.ssak; g.ssbk = c.ssbk; g.ssbn = c.ssbn; g.QM = c.QM; g.QN = c.QN; g.QK = c.QK;
        g.epi = c.epi; g.out_dtype = c.out_dtype; g.scale_dtype = c.scale_dtype; g.col_scalar = c.col_scalar; g.order = c.order;
        return gnk_launch_generic(op, fa, fb, g, st);
    }

    // split-K (planned above: int8 row/col-scaled product, fast layout, workspace offered and large enough)
    if (splits > 1) {
        {
            GemmParams p;
            memset(&p, 0, sizeof(p));
            p.A = (const char *)c.A; p.B = (const char *)c.B; p.C = c.ws;
            p.M = c.M; p.N = c.N; p.K = c.K; p.lda = la.ld; p.ldb = lb.ld; p.out_dtype = GNK_I32;
            p.kchunk = ((c.K / GNK_BK + splits - 1) / splits) * GNK_BK;
            if (first_steps > 0) {                    // uneven two-way split (256 x 256 kernel): split 0 = first_steps K-steps, split 1 = the rest
                p.kchunk0 = (int64_t)first_steps * GNK_BK;
                p.kchunk = c.K - p.kchunk0;
            }
            rc = variant == GNK_VARIANT_256_W4 ? gnk_launch_i8_v5(la.kmajor, lb.kmajor, EPI_PLAIN, p, st)
                 : variant == GNK_VARIANT_256 ? gnk_launch_i8_v0(la.kmajor, lb.kmajor, EPI_PLAIN, p, st) : gnk_launch_i8_v1(la.kmajor, lb.kmajor, EPI_PLAIN, p, st);
            if (rc) return rc;
            const int nsplit = first_steps > 0 ? 2 : (int)((c.K + p.kchunk - 1) / p.kchunk);
            const unsigned blocks = (unsigned)((c.M * c.N / 4 + 255) / 256);
#define GNK_REDUCE(OUT) hipLaunchKernelGGL((splitk_reduce_kernel<OUT>), dim3(blocks), dim3(256), 0, st, (const int *)c.ws, c.C, c.s0, c.s1, c.M, c.N, nsplit, c.scale_dtype, c.col_scalar, c.order)
            if (c.out_dtype == GNK_BF16) GNK_REDUCE(GNK_BF16);
            else if (c.out_dtype == GNK_F16) GNK_REDUCE(GNK_F16);
            else GNK_REDUCE(GNK_F32);
#undef GNK_REDUCE
            return gnk_check_launch("scaled_mm split-K reduction");
        }
    }

    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const char *)c.A; p.B = (const char *)c.B; p.C = c.C; p.s0 = c.s0; p.s1 = c.s1;
    p.M = c.M; p.N = c.N; p.K = c.K; p.lda = la.ld; p.ldb = lb.ld;
    p.ssam = c.ssam; p.ssak = c.ssak; p.ssbk = c.ssbk; p.ssbn = c.ssbn; p.QM = c.QM; p.QN = c.QN;
    p.out_dtype = c.out_dtype; p.scale_dtype = c.scale_dtype; p.col_scalar = c.col_scalar; p.order = c.order;
    p.debug = gnk_debug_flags_get();
    // tile-scaled rescale form: fused multiply-add for 16-bit outputs (their rounding hides the last-ulp difference), the interpreter's
    // three roundings for fp32 outputs (goldens g4 stay bit-exact); gnk_set_tile_rescale overrides either way
    p.fused_rescale = c.epi == EPI_TILE && (g_tile_rescale == 2 || (g_tile_rescale == 0 && c.out_dtype != GNK_F32)) ? 1 : 0;
    p.kgroup = c.epi == EPI_TILE && c.QK > GNK_BK ? (int)(c.QK / GNK_BK) : 1;
    if (c.epi == EPI_TILE && c.QK != GNK_BK) {   // Jetfire-style blocks / blocks of several K-steps: the 32-k MFMA instances of the 128 x 128 kernel
        if (op == OP_I8) return gnk_launch_tkb_i8((int)c.QK, la.kmajor, lb.kmajor, p, st);
        if (fa == F8_E4M3) return fb == F8_E4M3 ? gnk_launch_tkb_f8_00((int)c.QK, la.kmajor, lb.kmajor, p, st) : gnk_launch_tkb_f8_01((int)c.QK, la.kmajor, lb.kmajor, p, st);
        return fb == F8_E4M3 ? gnk_launch_tkb_f8_10((int)c.QK, la.kmajor, lb.kmajor, p, st) : gnk_launch_tkb_f8_11((int)c.QK, la.kmajor, lb.kmajor, p, st);
    }
    if (variant == GNK_VARIANT_64X128 && op == OP_F8) return gnk_launch_f8_v6(fa, fb, la.kmajor, lb.kmajor, c.epi, p, st);
    return pick_launcher(op, fa, fb, variant)(la.kmajor, lb.kmajor, c.epi, p, st);
}

}  // namespace

extern "C" int gnk_int8_mm(const int8_t *A, const int8_t *B, int32_t *C, int64_t M, int64_t N, int64_t K, int64_t sam,
                           int64_t sak, int64_t sbk, int64_t sbn, void *stream) {
    GemmCall c;
    memset(&c, 0, sizeof(c));
    c.A = A; c.a_dtype = GNK_I8; c.B = B; c.b_dtype = GNK_I8; c.C = C; c.out_dtype = GNK_I32; c.epi = EPI_PLAIN;
    c.M = M; c.N = N; c.K = K; c.sam = sam; c.sak = sak; c.sbk = sbk; c.sbn = sbn;
    return run_gemm("int8_mm", c, (hipStream_t)stream);
}

extern "C" int gnk_scaled_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
                             const void *row_scale, const void *col_scale, int scale_dtype, int col_scale_is_scalar,
                             int order, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbk,
                             int64_t sbn, void *stream) {
    GemmCall c;
    memset(&c, 0, sizeof(c));
    c.A = A; c.a_dtype = a_dtype; c.B = B; c.b_dtype = b_dtype; c.C = C; c.out_dtype = out_dtype;
    c.s0 = row_scale; c.s1 = col_scale; c.scale_dtype = scale_dtype; c.col_scalar = col_scale_is_scalar != 0;
    c.order = order != 0; c.epi = EPI_ROWCOL;
    c.M = M; c.N = N; c.K = K; c.sam = sam; c.sak = sak; c.sbk = sbk; c.sbn = sbn;
    return run_gemm("scaled_mm", c, (hipStream_t)stream);
}

extern "C" int64_t gnk_scaled_mm_workspace_bytes(int a_dtype, int b_dtype, int64_t M, int64_t N, int64_t K) {
    if (a_dtype != GNK_I8 || b_dtype != GNK_I8 || M <= 0 || N <= 0) return 0;
    if (K < GNK_BK || K % GNK_BK != 0 || N % 4 != 0) return 0;
    // layout / alignment are re-checked at launch, where the plan is the one of the kernel that takes the call: offer room for either
    const GemmPlan plan = plan_gemm(M, N, K, true, true, true, false), plan4 = plan_gemm(M, N, K, true, true, true, K % 256 == 0);
    const int splits = plan.splits > plan4.splits ? plan.splits : plan4.splits;
    return splits > 1 ? (int64_t)splits * M * N * 4 : 0;
}

extern "C" int gnk_scaled_mm_ws(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
                                const void *row_scale, const void *col_scale, int scale_dtype, int col_scale_is_scalar,
                                int order, int64_t M, int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbk,
                                int64_t sbn, void *workspace, int64_t workspace_bytes, void *stream) {
    GemmCall c;
    memset(&c, 0, sizeof(c));
    c.A = A; c.a_dtype = a_dtype; c.B = B; c.b_dtype = b_dtype; c.C = C; c.out_dtype = out_dtype;
    c.s0 = row_scale; c.s1 = col_scale; c.scale_dtype = scale_dtype; c.col_scalar = col_scale_is_scalar != 0;
    c.order = order != 0; c.epi = EPI_ROWCOL;
    c.M = M; c.N = N; c.K = K; c.sam = sam; c.sak = sak; c.sbk = sbk; c.sbn = sbn;
    c.ws = workspace; c.ws_bytes = workspace ? workspace_bytes : 0;
    return run_gemm("scaled_mm", c, (hipStream_t)stream);
}

extern "C" int gnk_tile_scaled_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype,
                                  const void *sA, const void *sB, int scale_dtype, int64_t M, int64_t N, int64_t K,
                                  int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ssam, int64_t ssak,
                                  int64_t ssbk, int64_t ssbn, int64_t QM, int64_t QN, int64_t QK, void *stream) {
    GemmCall c;
    memset(&c, 0, sizeof(c));
    c.A = A; c.a_dtype = a_dtype; c.B = B; c.b_dtype = b_dtype; c.C = C; c.out_dtype = out_dtype;
    c.s0 = sA; c.s1 = sB; c.scale_dtype = scale_dtype; c.epi = EPI_TILE;
    c.M = M; c.N = N; c.K = K; c.sam = sam; c.sak = sak; c.sbk = sbk; c.sbn = sbn;
    c.ssam = ssam; c.ssak = ssak; c.ssbk = ssbk; c.ssbn = ssbn; c.QM = QM; c.QN = QN; c.QK = QK;
    return run_gemm("tile_scaled_mm", c, (hipStream_t)stream);
}

extern "C" int gnk_fp8_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype, int64_t M,
                          int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, void *stream) {
    if (!gnk_is_f8(a_dtype) || !gnk_is_f8(b_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "fp8_mm: operands must be fp8 (got dtype codes %d, %d)", a_dtype, b_dtype);
    GemmCall c;
    memset(&c, 0, sizeof(c));
    c.A = A; c.a_dtype = a_dtype; c.B = B; c.b_dtype = b_dtype; c.C = C; c.out_dtype = out_dtype; c.epi = EPI_PLAIN;
    c.M = M; c.N = N; c.K = K; c.sam = sam; c.sak = sak; c.sbk = sbk; c.sbn = sbn;
    return run_gemm("fp8_mm", c, (hipStream_t)stream);
}
