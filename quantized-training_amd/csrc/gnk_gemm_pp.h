// gnk_gemm_pp.h -- 256 x 256 "ping-pong" MFMA GEMM for 8-bit operands on gfx950.
//
// Same math, LDS images, staging and epilogue as gnk_gemm_kernel.h; what changes is the main loop.
// A workgroup is 8 waves = two groups of four (group g owns output rows 128g .. 128g+127, one wave per
// SIMD in each group), so every SIMD hosts one wave of each group.  A K-step (128 k) is cut into
// 8 phases separated by s_barrier; in every phase ONE group issues a cluster of MFMAs for one
// 64 x 32 quadrant of its 128 x 64 wave tile while the OTHER group reads the fragments of its next
// quadrant from LDS and issues the LDS-DMA loads of a later K-step:
//
//     phase      0      1      2      3      4      5      6      7      (one K-step = 8 phases)
//     group 0   MFMA   LOAD   MFMA   LOAD   MFMA   LOAD   MFMA   LOAD
//     group 1   LOAD   MFMA   LOAD   MFMA   LOAD   MFMA   LOAD   MFMA
//
// Both groups run the SAME program; group 1 is delayed by one barrier at entry (group 0 pays it back
// at exit).  Each SIMD's matrix pipe therefore always has exactly one wave feeding it, and the
// LDS reads, DMA issue and waits of the other wave are hidden behind those MFMAs.
//
// Per-group program for K-step t (buffer t & 1), quadrant order (0,0) (0,1) (1,1) (1,0) so that
// consecutive quadrants share an operand sub-tile:
//     p0 MFMA q(0,0) | p1 read B1; DMA B-pieces of tile t+1 | p2 MFMA q(0,1) | p3 read A1
//     p4 MFMA q(1,1) | p5 read B0; vmcnt(0): tile t+1 landed | p6 MFMA q(1,0)
//     p7 read A0,B0 of tile t+1; DMA A-pieces of tile t+2 into tile t's buffer (K-major x K-major: in p5, behind the wait)
// Buffer safety (2 LDS buffers): tile t is last read in p5 of either group (reads retired by
// lgkmcnt(0) before the phase's barrier), and the first DMA into its buffer is issued in p7 of group 0,
// one barrier later.  Tile t+1 is complete when every wave has passed its own vmcnt(0) in p5 and
// the barrier that follows; its first reader is p7.
#pragma once
#include <atomic>
#include "gnk_gemm_kernel.h"

namespace gnk_gemm {

template <class Cfg> struct PP {
    static_assert(Cfg::BM == 256 && Cfg::BN == 256 && Cfg::WM == 2 && Cfg::WN == 4, "ping-pong kernel is 256x256, 2x4 waves");
    static_assert(Cfg::EPI != EPI_TILE, "tile-scaled epilogue uses the two-buffer kernel");
    typedef typename AccT<Cfg::OP>::type acc_t;
    typedef Stager<Cfg::A_KMAJOR, 256, 8> StA;
    typedef Stager<Cfg::B_KMAJOR, 256, 8> StB;
    static constexpr bool DMA_P5 = Cfg::A_KMAJOR && Cfg::B_KMAJOR;

    // Fragments are kept as 8-register tuples (k-substep 0 in elements 0-3, k-substep 1 in 4-7): exactly the
    // operand of the K=128 fp8 MFMA, and two sub-tuples for the K=64 int8 MFMA, so that no copy is needed
    // between the phase that reads them from LDS and the phase that feeds them to the matrix pipe.
    static __device__ __forceinline__ v8i cat(const v4i &lo, const v4i &hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

    template <int QM> static __device__ __forceinline__ void load_a(const char *As, v8i (&fa)[4], int wm, int lane) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            fa[i] = cat(load_frag<Cfg::A_KMAJOR, 256>(As, wm * 128 + QM * 64 + 16 * i, 0, lane),
                        load_frag<Cfg::A_KMAJOR, 256>(As, wm * 128 + QM * 64 + 16 * i, 1, lane));
    }
    template <int QN> static __device__ __forceinline__ void load_b(const char *Bs, v8i (&fb)[2], int wn, int lane) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            fb[j] = cat(load_frag<Cfg::B_KMAJOR, 256>(Bs, wn * 64 + QN * 32 + 16 * j, 0, lane),
                        load_frag<Cfg::B_KMAJOR, 256>(Bs, wn * 64 + QN * 32 + 16 * j, 1, lane));
    }
    template <int QM, int QN>
    static __device__ __forceinline__ void mfma_q(acc_t (&acc)[8][4], const v8i (&fa)[4], const v8i (&fb)[2]) {
        if constexpr (Cfg::OP == OP_I8) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const v4i a = ks == 0 ? __builtin_shufflevector(fa[i], fa[i], 0, 1, 2, 3) : __builtin_shufflevector(fa[i], fa[i], 4, 5, 6, 7);
                        const v4i b = ks == 0 ? __builtin_shufflevector(fb[j], fb[j], 0, 1, 2, 3) : __builtin_shufflevector(fb[j], fb[j], 4, 5, 6, 7);
                        acc[QM * 4 + i][QN * 2 + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, acc[QM * 4 + i][QN * 2 + j], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[QM * 4 + i][QN * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        fb[j], fa[i], acc[QM * 4 + i][QN * 2 + j], Cfg::FB, Cfg::FA, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // E8M0 127 = 2^0
        }
        // Pin the cluster to this phase: an MFMA is a register-only op, so nothing else stops LLVM from
        // sinking it past the barriers to the loop latch (it did, for the one-MFMA-per-accumulator fp8 form).
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[QM * 4 + i][QN * 2 + j]));
    }

    static __device__ __forceinline__ void end_mfma_phase() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    static __device__ __forceinline__ void end_load_phase() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers; LDS reads retired
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // One K-step.  BUF is the LDS buffer of tile t; tiles t+1 / t+2 go to BUF^1 / BUF.
    template <int BUF>
    static __device__ __forceinline__ void kstep(char *smem, int t, int nk, acc_t (&acc)[8][4], v8i (&fa)[4], v8i (&fb)[2],
                                                 StA &sa, StB &sb, int wave, int wm, int wn, int lane, int dbg) {
        char *const As = smem + BUF * Cfg::A_BYTES, *const An = smem + (BUF ^ 1) * Cfg::A_BYTES;
        char *const Bs = smem + 2 * Cfg::A_BYTES + BUF * Cfg::B_BYTES, *const Bn = smem + 2 * Cfg::A_BYTES + (BUF ^ 1) * Cfg::B_BYTES;
        const bool dma = !(dbg & 1), mma = !(dbg & 2), rd = !(dbg & 4);   // timing ablations; all true in production
        if (mma) mfma_q<0, 0>(acc, fa, fb);              // p0
        end_mfma_phase();
        if (rd) load_b<1>(Bs, fb, wn, lane);             // p1
        if (t + 1 < nk && dma) sb.issue(Bn, wave);       //    second half of tile t+1 (first half: p7 of t-1, or the prologue)
        end_load_phase();
        if (mma) mfma_q<0, 1>(acc, fa, fb);              // p2
        end_mfma_phase();
        if (rd) load_a<1>(As, fa, wm, lane);             // p3
        end_load_phase();
        if (mma) mfma_q<1, 1>(acc, fa, fb);              // p4
        end_mfma_phase();
        if (rd) load_b<0>(Bs, fb, wn, lane);             // p5
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); //    this wave's pieces of tile t+1 have landed
        // A-half of tile t+2 into tile t's A buffer (last read in p3 of either group).  Both operands K-major: issued here, behind the
        // wait, where the phase carries 4 fragment reads (p7 carries 12): 1-2 % at 4096^3; the M/N-major forms showed no gain and keep p7
        if (DMA_P5 && t + 2 < nk && dma) sa.issue(As, wave);
        end_load_phase();
        if (mma) mfma_q<1, 0>(acc, fa, fb);              // p6
        end_mfma_phase();
        if (t + 1 < nk && rd) {                          // p7
            load_a<0>(An, fa, wm, lane);
            load_b<0>(Bn, fb, wn, lane);
        }
        if (!DMA_P5 && t + 2 < nk && dma) sa.issue(As, wave);   //    first half of tile t+2 into tile t's buffer
        end_load_phase();
    }
};

template <class Cfg>
__global__ void __launch_bounds__(512) gemm_pp_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef PP<Cfg> K;
    typedef typename K::acc_t acc_t;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

    ScalePrefetch<256, 256, 512, Cfg::EPI> spf;
    spf.issue(p, m0, n0);

    typename K::StA sa;
    typename K::StB sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);

    acc_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc_t){0, 0, 0, 0};
    v8i fa[4], fb[2];

    const int nk = (int)(p.K / GNK_BK);
    char *const A0 = smem, *const B0 = smem + 2 * Cfg::A_BYTES;
    // prologue: tile 0 in full and the A half of tile 1 (its B half goes out in p1 of K-step 0, as in
    // steady state), then the first quadrant's fragments
    sa.issue(A0, wave);
    sb.issue(B0, wave);
    if (nk > 1) sa.issue(A0 + Cfg::A_BYTES, wave);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 (the 8 oldest loads) has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    K::template load_a<0>(A0, fa, wm, lane);
    K::template load_b<0>(B0, fb, wn, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // Static priority for the later-dispatched half of the workgroup (waves 4-7 lose every issue arbitration on their SIMD otherwise),
    // set once: measured 1-2 % ahead of raising the priority around every MFMA cluster (MI355X_MICROARCH.md, "Two waves per SIMD", item 4).
    if (wm == 1) __builtin_amdgcn_s_setprio(1);
    if (wm == 1) __builtin_amdgcn_s_barrier();           // stagger: group 1 runs one phase behind group 0
    __builtin_amdgcn_sched_barrier(0);

    const int dbg = GNK_DEBUG_BITS(p.debug);
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        K::template kstep<0>(smem, t, nk, acc, fa, fb, sa, sb, wave, wm, wn, lane, dbg);
        K::template kstep<1>(smem, t + 1, nk, acc, fa, fb, sa, sb, wave, wm, wn, lane, dbg);
    }
    if (t < nk) K::template kstep<0>(smem, t, nk, acc, fa, fb, sa, sb, wave, wm, wn, lane, dbg);
    if (wm == 0) __builtin_amdgcn_s_barrier();           // pay the stagger back so barrier counts match
    __builtin_amdgcn_sched_barrier(0);
    spf.park(p, smem + Cfg::LDS_BYTES);
    __syncthreads();                                     // scales parked in LDS are visible to every wave
    if (dbg & 8) return;                                 // timing ablation: no epilogue at all

    // all LDS tile reads retired at the last barrier; each wave stages its result through its own LDS slice
    v4f(*no_accf)[1] = nullptr;
    run_epilogue<Cfg>(p, acc, no_accf, smem, wave, lane, m0 + wm * 128, n0 + wn * 64,
                      Cfg::EPI == EPI_ROWCOL ? (const float *)(smem + Cfg::LDS_BYTES) : nullptr, wm * 128, wn * 64);
}

template <class Cfg> int launch_pp(const GemmParams &p0, hipStream_t st) {
    constexpr int LDS_TOTAL = Cfg::LDS_BYTES + ScalePrefetch<256, 256, 512, Cfg::EPI>::BYTES;
    GemmParams p = p0;
    p.grid_m = (int)((p.M + 255) / 256);
    p.grid_n = (int)((p.N + 255) / 256);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "gemm: no current HIP device");
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void *)gemm_pp_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return gnk_fail(GNK_ERR_LAUNCH, "gemm: cannot reserve %d bytes of LDS", LDS_TOTAL);
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((gemm_pp_kernel<Cfg>), dim3((unsigned)(p.grid_m * p.grid_n)), dim3(512), LDS_TOTAL, st, p);
    return gnk_check_launch("gemm (ping-pong)");
}

}  // namespace gnk_gemm
