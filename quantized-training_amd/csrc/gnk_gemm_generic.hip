// gnk_gemm_generic.hip -- fallback GEMM for anything the LDS-DMA fast path does not take: any
// strides (including both non-unit), any M/N/K (K tails, K < 128), unaligned bases, and tile
// scales with arbitrary block sizes (e.g. Jetfire 32x32 blocks, triton_mm.py:410-441).
//
// 64 x 64 output tile per 256-thread workgroup, K advanced in 32-element chunks that never cross
// a scale block; operands are gathered element-wise (bounds-checked, zero filled) into K-major
// LDS images and fed to the 8-byte-per-lane MFMAs (v_mfma_i32_16x16x32_i8 /
// v_mfma_f32_16x16x32_{fp8,bf8}_{fp8,bf8}).  Correctness first; the fast path carries the load.
#include "gnk_gemm_dispatch.h"

namespace {

constexpr int GB = 64;       // tile edge
constexpr int GK = 32;       // K chunk
constexpr int GP = 40;       // LDS row pitch in bytes (8-byte aligned, breaks the 32-byte stride)

template <int OP, int FA, int FB>
__global__ void __launch_bounds__(256) gemm_generic_kernel(const GenericParams p) {
    __shared__ __attribute__((aligned(16))) char As[GB * GP];
    __shared__ __attribute__((aligned(16))) char Bs[GB * GP];
    __shared__ float sAs[GB], sBs[GB];        // tile scales of the current K-block: rows m0 .. m0 + 63, columns n0 .. n0 + 63
    typedef typename gnk_gemm::AccT<OP>::type acc_t;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GB, n0 = (int64_t)blockIdx.x * GB;

    // loader role: row lr of the tile, 8 consecutive k starting at lk
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 8;
    const int64_t am = m0 + lr, bn = n0 + lr;
    // loader role for an operand whose M / N dim is the contiguous one: k = ck of the chunk, 8 consecutive rows starting at cr
    const int ck = threadIdx.x >> 3, cr = (threadIdx.x & 7) * 8;
    const bool a_rowc = p.sam == 1 && p.sak != 1, b_rowc = p.sbn == 1 && p.sbk != 1;

    acc_t part[2][2];
    v4f accf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            part[i][j] = (acc_t){0, 0, 0, 0};
            accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
        }

    const int64_t mrow = m0 + wm * 32 + (lane & 15);
    const int64_t ncol = n0 + wn * 32 + 4 * (lane >> 4);
    const int64_t QK = p.epi == EPI_TILE ? p.QK : (p.K > 0 ? p.K : 1);

    for (int64_t kb0 = 0, kb = 0; kb0 < p.K; kb0 += QK, ++kb) {
        const int64_t kend = kb0 + QK < p.K ? kb0 + QK : p.K;
        for (int64_t k0 = kb0; k0 < kend; k0 += GK) {
            // Gather the chunk (64 rows x 32 k per operand) into the K-major LDS images.  An operand whose K dim is contiguous is read as
            // 8 consecutive k of one row per thread; one whose M / N dim is contiguous as 8 consecutive rows of one k per thread (coalesced
            // along the contiguous dim, scattered into 8 LDS rows); either way ONE 8-byte load when the group is whole, in bounds and 8-byte
            // aligned, else element by element (bounds-checked, zero filled) -- which is also the path of doubly strided operands.
            // (The first version read every operand byte by byte: 22 TOP/s at 4096^3.)
            unsigned long long pa = 0, pb = 0;
            const int64_t kk = k0 + lk;              // K-contiguous role: row lr, k = kk .. kk + 7
            const int64_t kc = k0 + ck;              // row-contiguous role: k = kc, rows cr .. cr + 7
            if (a_rowc) {
                const int64_t m = m0 + cr;
                const char *src = p.A + m * p.sam + kc * p.sak;
                if (kc < kend) {
                    if (m + 8 <= p.M && ((uintptr_t)src & 7) == 0) pa = *(const unsigned long long *)src;
                    else
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (m + e < p.M) pa |= (unsigned long long)(unsigned char)src[e] << (8 * e);
                }
            } else {
                const char *src = p.A + am * p.sam + kk * p.sak;
                if (am < p.M) {
                    if (p.sak == 1 && kk + 8 <= kend && ((uintptr_t)src & 7) == 0) pa = *(const unsigned long long *)src;
                    else
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (kk + e < kend) pa |= (unsigned long long)(unsigned char)src[e * p.sak] << (8 * e);
                }
            }
            if (b_rowc) {
                const int64_t n = n0 + cr;
                const char *src = p.B + kc * p.sbk + n * p.sbn;
                if (kc < kend) {
                    if (n + 8 <= p.N && ((uintptr_t)src & 7) == 0) pb = *(const unsigned long long *)src;
                    else
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (n + e < p.N) pb |= (unsigned long long)(unsigned char)src[e] << (8 * e);
                }
            } else {
                const char *src = p.B + kk * p.sbk + bn * p.sbn;
                if (bn < p.N) {
                    if (p.sbk == 1 && kk + 8 <= kend && ((uintptr_t)src & 7) == 0) pb = *(const unsigned long long *)src;
                    else
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (kk + e < kend) pb |= (unsigned long long)(unsigned char)src[e * p.sbk] << (8 * e);
                }
            }
            __syncthreads();   // previous chunk's fragments (and the previous K-block's scales) have been read
            if (p.epi == EPI_TILE && k0 == kb0 && threadIdx.x < 2 * GB) {
                // this K-block's scales, one per thread (per-lane loads in the rescale, 10 dependent global loads per block and lane,
                // held the tile-scaled fallback at 16 TOP/s for 16-k blocks)
                const int t = threadIdx.x & (GB - 1);
                if (threadIdx.x < GB) {
                    int64_t m = m0 + t;
                    m = m < p.M ? m : p.M - 1;
                    sAs[t] = gnk_load_scalar_rt(p.s0, p.scale_dtype, (m / p.QM) * p.ssam + kb * p.ssak);
                } else {
                    int64_t n = n0 + t;
                    n = n < p.N ? n : p.N - 1;
                    sBs[t] = gnk_load_scalar_rt(p.s1, p.scale_dtype, kb * p.ssbk + (n / p.QN) * p.ssbn);
                }
            }
            if (a_rowc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) As[(cr + e) * GP + ck] = (char)(pa >> (8 * e));
            } else {
                *(unsigned long long *)(As + lr * GP + lk) = pa;
            }
            if (b_rowc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) Bs[(cr + e) * GP + ck] = (char)(pb >> (8 * e));
            } else {
                *(unsigned long long *)(Bs + lr * GP + lk) = pb;
            }
            __syncthreads();
            long fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const long *)(As + (wm * 32 + 16 * i + (lane & 15)) * GP + 8 * (lane >> 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const long *)(Bs + (wn * 32 + 16 * j + (lane & 15)) * GP + 8 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // MFMA "A" operand <- matrix B: lane ends up with m = lane & 15, n = 4 * (lane >> 4) + r
                    if constexpr (OP == OP_I8) part[i][j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(fb[j], fa[i], part[i][j], 0, 0, 0);
                    else if constexpr (FB == F8_E4M3 && FA == F8_E4M3) part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j], fa[i], part[i][j], 0, 0, 0);
                    else if constexpr (FB == F8_E4M3 && FA == F8_E5M2) part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[j], fa[i], part[i][j], 0, 0, 0);
                    else if constexpr (FB == F8_E5M2 && FA == F8_E4M3) part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(fb[j], fa[i], part[i][j], 0, 0, 0);
                    else part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(fb[j], fa[i], part[i][j], 0, 0, 0);
                }
        }
        if (p.epi == EPI_TILE) {
            // acc_f32 += (f32(partial) * sA[m/QM, kb]) * sB[kb, n/QN], separate roundings (triton_mm.py:327-331)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float sa = sAs[wm * 32 + 16 * i + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sb = sBs[wn * 32 + 4 * (lane >> 4) + 16 * j + r];
                        float a;
                        if constexpr (OP == OP_I8) a = (float)part[i][j][r];
                        else a = part[i][j][r];
                        accf[i][j][r] = __fadd_rn(accf[i][j][r], __fmul_rn(__fmul_rn(a, sa), sb));
                    }
                    part[i][j] = (acc_t){0, 0, 0, 0};
                }
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = mrow + 16 * i;
        if (m >= p.M) continue;
        const float rs = p.epi == EPI_ROWCOL ? gnk_load_scalar_rt(p.s0, p.scale_dtype, m) : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = ncol + 16 * j + r;
                if (n >= p.N) continue;
                const int64_t idx = m * p.N + n;
                if (p.epi == EPI_PLAIN) {
                    if constexpr (OP == OP_I8) ((int *)p.C)[idx] = part[i][j][r];
                    else gnk_store_scalar_rt(p.C, p.out_dtype, idx, part[i][j][r]);
                } else if (p.epi == EPI_ROWCOL) {
                    const float cs = gnk_load_scalar_rt(p.s1, p.scale_dtype, p.col_scalar ? 0 : n);
                    float a;
                    if constexpr (OP == OP_I8) a = (float)part[i][j][r];
                    else a = part[i][j][r];
                    const float v = p.order == 0 ? __fmul_rn(__fmul_rn(a, rs), cs) : __fmul_rn(__fmul_rn(a, cs), rs);
                    gnk_store_scalar_rt(p.C, p.out_dtype, idx, v);
                } else {
                    gnk_store_scalar_rt(p.C, p.out_dtype, idx, accf[i][j][r]);
                }
            }
    }
}

template <int OP, int FA, int FB> int launch_generic(const GenericParams &p, hipStream_t st) {
    const dim3 grid((unsigned)((p.N + GB - 1) / GB), (unsigned)((p.M + GB - 1) / GB));
    if (grid.y > 65535u) return gnk_fail(GNK_ERR_UNSUPPORTED, "generic gemm: M = %lld exceeds the fallback kernel's grid", (long long)p.M);
    hipLaunchKernelGGL((gemm_generic_kernel<OP, FA, FB>), grid, dim3(256), 0, st, p);
    return gnk_check_launch("gemm (generic)");
}

}  // namespace

int gnk_launch_generic(int op, int fa, int fb, const GenericParams &p, hipStream_t st) {
    if (op == OP_I8) return launch_generic<OP_I8, 0, 0>(p, st);
    if (fa == F8_E4M3) return fb == F8_E4M3 ? launch_generic<OP_F8, F8_E4M3, F8_E4M3>(p, st) : launch_generic<OP_F8, F8_E4M3, F8_E5M2>(p, st);
    return fb == F8_E4M3 ? launch_generic<OP_F8, F8_E5M2, F8_E4M3>(p, st) : launch_generic<OP_F8, F8_E5M2, F8_E5M2>(p, st);
}
