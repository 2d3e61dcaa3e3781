#include <atomic>
// gnk_gemm_mx.hip -- MXFP4 and MXFP8 GEMMs on CDNA4's block-scaled MFMA.
//
//   C(M x N, bf16) = cast( sum over 32-k blocks kb of  (A_kb . B_kb) * 2^(sA[m,kb]-127) * 2^(sB[n,kb]-127)  (+ bias[n]) )
//
// Replaces: kernels/gn_kernels/cutlass_mm.py:80-86 (mxfp4_mm) -> kernels/csrc/cutlass_sm120a_fp4.cu:42-136, a CUTLASS
// block-scaled GEMM for sm120a.  A is (M, K/2) bytes and B.T is (N, K/2) bytes, two e2m1 values per byte (even k in the
// low nibble, kernels/gn_kernels/utils.py:52-80); scales are E8M0, one per row and 32 consecutive k.
//
// MI355X mapping: v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 operands consumes, per lane l = 16 g + r, row r's 32 values
// k = 32 g .. 32 g + 31 (16 bytes) and ONE E8M0 scale byte for exactly those 32 values (tools/probe_mx.hip checks this
// against a CPU MX product): an MX block is a lane.  A row of 128 bytes (256 k) is therefore the int8 kernels' 128-byte
// K-step unchanged -- same LDS-DMA staging, same XOR-swizzled K-major image, same 16-byte fragment reads, 2 MFMAs per tile
// and step -- at twice the arithmetic per byte.  Scales: the workgroup stages the scale bytes of its rows for 8 K-steps
// at a time in LDS (coalesced 8-byte loads), every lane reads the 8 bytes of each of its row tiles per step from there
// and shifts its own block's byte to byte 0 (op_sel 0).
//
// MXFP8 (the consumer of quantize_mx's e4m3 / e5m2 output, kernels/gn_kernels/utils.py:83-107; the reference has no such GEMM): the same
// instruction with FP8 operands takes 32 bytes per lane -- k = 16 g .. 16 g + 15 and k = 64 + 16 g .. + 15 of row r, exactly the two
// 16-byte chunks the fp8 kernels' fragment reads deliver -- and the scale byte held by lane 16 g + r scales k = 32 g .. 32 g + 31
// of row r (tools/probe_mx8.hip: blocks = {(g0,h0),(g1,h0)}, {(g2,h0),(g3,h0)}, {(g0,h1),(g1,h1)}, {(g2,h1),(g3,h1)}): natural MX
// blocks, one MFMA per tile and 128-byte step, the lane's scale = sA[row][4 t + g].
#include "gnk_gemm_dispatch.h"

#include <string.h>
#include <type_traits>

namespace {

typedef GemmCfg<128, 128, 2, 2, true, true, OP_F8, 0, 0, EPI_PLAIN> MxCfg;   // fp32 accumulators, plain epilogue maps

struct MxParams {
    GemmParams g;                 // A, B (packed bytes), C, M, N, K = packed bytes per row (k / 2), lda / ldb in bytes
    const unsigned char *sa;      // (M, k / 32) E8M0, row stride ldsa bytes
    const unsigned char *sb;      // (N, k / 32)
    int64_t ldsa, ldsb;
    const void *bias;             // (N) or nullptr
    int bias_dtype;
};

constexpr int FMT_FP4 = 4;        // cbsz / blgp code of e2m1
constexpr int MX_LDS = MxCfg::LDS_BYTES + 2 * 128 * 64;   // operand buffers + the scale panel: 80 KiB, two workgroups per CU

// E8M0 scale byte -> fp32 as the OCP MX spec (and the MFMA's scale input) reads it: 2^(s - 127); s == 0 is 2^-127 (an fp32 subnormal, not 0)
// and s == 255 is NaN.
__device__ __forceinline__ float gnk_e8m0_to_f32(unsigned char s) {
    return s == 255 ? __uint_as_float(0x7fc00000u) : s == 0 ? __uint_as_float(0x00400000u) : __uint_as_float((unsigned)s << 23);
}

// BPS = MX blocks per 128-byte K-step: 8 = FP4 (256 k, two MFMAs per tile and step, FA = FB = 4), 4 = FP8 (128 k, one MFMA; FA / FB =
// 0 e4m3, 1 e5m2).  The scale panel holds 64 bytes per operand row either way: 8 (FP4) or 16 (FP8) K-steps.
template <int BPS, int FA, int FB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) mx_kernel(const MxParams mp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef MxCfg Cfg;
    constexpr int MT = Cfg::MT, NTL = Cfg::NTL;
    const GemmParams &p = mp.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    int tm, tn;
    gnk_gemm::tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * Cfg::BM, n0 = (int64_t)tn * Cfg::BN;
    char *const As0 = smem, *const Bs0 = smem + 2 * Cfg::A_BYTES;

    gnk_gemm::Stager<true, Cfg::BM, Cfg::NW> sa;
    gnk_gemm::Stager<true, Cfg::BN, Cfg::NW> sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);

    v4f acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    // ---- scales.  Each lane needs, per K-step, 8 scale bytes of every operand row it feeds (row = lane & 15 of each 16-row
    // tile; its own MX block inside an MFMA = lane >> 4).  Fetching them per lane from global memory (16 rows x 8 B per
    // load instruction) took as long as the MFMAs (90 vs 46 us at 4096^3), so the workgroup stages a PANEL -- the scale
    // bytes of its 128 + 128 rows for 8 K-steps, 64 B per row, 16 KiB -- in LDS behind the operand buffers with coalesced
    // 8-byte loads, once per 8 steps, and every lane reads its 8 bytes per row tile and step from there.
    constexpr int PROW = 64, PSTEPS = PROW / BPS;          // bytes per panel row, K-steps per panel
    char *const panel = smem + Cfg::LDS_BYTES;             // [2 operands][128 rows][PROW]
    const int r = lane & 15, sh = 8 * (lane >> 4);
    const int nk = (int)(p.K / GNK_BK);                    // 128 packed bytes = 256 k = 8 MX blocks per step
    const int64_t row_bytes = p.K * BPS / GNK_BK;          // scale bytes per row (k / 32)
    // this thread's 8 panel chunks (8 B each): chunk c = tid + 256 u -> operand c >> 10, row (c >> 3) & 127, part c & 7
    const unsigned char *csrc[8];
    int cpart[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = (int)threadIdx.x + 256 * u, op = c >> 10, row = (c >> 3) & 127;
        cpart[u] = c & 7;
        int64_t g = (op ? n0 : m0) + row;
        const int64_t last = (op ? p.N : p.M) - 1;
        g = g < last ? g : last;
        csrc[u] = (op ? mp.sb + g * mp.ldsb : mp.sa + g * mp.ldsa) + 8 * cpart[u];
    }
    v2i pnl[8];
    auto panel_fetch = [&](int pn) {                        // global -> registers (bytes past the row end are never used)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            pnl[u] = (v2i){0, 0};
            if ((int64_t)PROW * pn + 8 * cpart[u] < row_bytes) pnl[u] = *(const v2i *)(csrc[u] + (int64_t)PROW * pn);
        }
    };
    auto panel_store = [&]() {                              // registers -> LDS
#pragma unroll
        for (int u = 0; u < 8; ++u) *(GNK_LDS v2i *)(GNK_LDS char *)(panel + ((int)threadIdx.x + 256 * u) * 8) = pnl[u];
    };
    if (nk > 0) {
        panel_fetch(0);
        sa.issue(As0, wave);
        sb.issue(Bs0, wave);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the 8 panel loads are older than the 8 LDS-DMA pieces
        panel_store();
    }
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int ts = t % PSTEPS;
        if (ts == 0 && t > 0) {                             // the panel of steps t .. t + 7 was fetched during step t - 1
            panel_store();
            __syncthreads();
        }
        // this step's scale bytes: block lane >> 4 of each 128-k group, moved to byte 0 (op_sel 0)
        int sca[MT][2], scb[NTL][2];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const char *src = panel + (wm * Cfg::WTM + 16 * i + r) * PROW + BPS * ts;
            if constexpr (BPS == 8) {
                const v2i w = *(const GNK_LDS v2i *)(const GNK_LDS char *)src;
                sca[i][0] = (int)(((unsigned)w[0] >> sh) & 0xffu);
                sca[i][1] = (int)(((unsigned)w[1] >> sh) & 0xffu);
            } else {
                sca[i][0] = sca[i][1] = (int)((*(const GNK_LDS unsigned *)(const GNK_LDS char *)src >> sh) & 0xffu);
            }
        }
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const char *src = panel + (128 + wn * Cfg::WTN + 16 * j + r) * PROW + BPS * ts;
            if constexpr (BPS == 8) {
                const v2i w = *(const GNK_LDS v2i *)(const GNK_LDS char *)src;
                scb[j][0] = (int)(((unsigned)w[0] >> sh) & 0xffu);
                scb[j][1] = (int)(((unsigned)w[1] >> sh) & 0xffu);
            } else {
                scb[j][0] = scb[j][1] = (int)((*(const GNK_LDS unsigned *)(const GNK_LDS char *)src >> sh) & 0xffu);
            }
        }
        if (ts == PSTEPS - 1 && t + 1 < nk && !(GNK_DEBUG_BITS(p.debug) & 128)) panel_fetch((t + 1) / PSTEPS);   // debug flag 128: timing only
        const int cur = t & 1;
        if (t + 1 < nk) {
            sa.issue(As0 + (cur ^ 1) * Cfg::A_BYTES, wave);
            sb.issue(Bs0 + (cur ^ 1) * Cfg::B_BYTES, wave);
        }
        const char *As = As0 + cur * Cfg::A_BYTES, *Bs = Bs0 + cur * Cfg::B_BYTES;
        if constexpr (BPS == 4) {
            // FP8: one MFMA per tile over the whole 128-byte step; fragment = chunks g and 4 + g (k = 16 g .. and 64 + 16 g ..)
            v8i fa[MT], fb[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const v4i lo = gnk_gemm::load_frag<true, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, 0, lane), hi = gnk_gemm::load_frag<true, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, 1, lane);
                fb[j] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const v4i lo = gnk_gemm::load_frag<true, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, 0, lane), hi = gnk_gemm::load_frag<true, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, 1, lane);
                fa[i] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)   // the first MFMA operand is the B fragment: its format code is FB
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], FB, FA, 0, scb[j][0], 0, sca[i][0]);
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8i fa[MT], fb[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const v4i f = gnk_gemm::load_frag<true, Cfg::BN>(Bs, wn * Cfg::WTN + 16 * j, ks, lane);
                fb[j] = (v8i){f[0], f[1], f[2], f[3], 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const v4i f = gnk_gemm::load_frag<true, Cfg::BM>(As, wm * Cfg::WTM + 16 * i, ks, lane);
                fa[i] = (v8i){f[0], f[1], f[2], f[3], 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)   // matrix B feeds the MFMA's A port (a lane then owns 4 consecutive n of one m)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], FA, FB, 0, scb[j][ks], 0, sca[i][ks]);
        }
    }

    // bias: fp32 add before the cast (cutlass_sm120a_fp4.cu:56-57: Add in the accumulator type, then Cast)
    if (mp.bias) {
        const int64_t ncol = n0 + wn * Cfg::WTN + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int64_t n = ncol + 16 * j + q;
                n = n < p.N ? n : p.N - 1;
                const float b = gnk_load_scalar_rt(mp.bias, mp.bias_dtype, n);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][j][q] = __fadd_rn(acc[i][j][q], b);
            }
    }
    __syncthreads();   // every wave is done reading the operand tiles: LDS is free for the result staging
    v4f(*no_accf)[1] = nullptr;
    gnk_gemm::run_epilogue<Cfg>(p, acc, no_accf, smem, wave, lane, m0 + wm * Cfg::WTM, n0 + wn * Cfg::WTN);
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves, four 64-byte stages, two-phase ping-pong: the int8 program of gnk_gemm_pp2.h with the scaled FP4
// MFMA.  One 64-byte stage row is 128 k = exactly one v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile, and its four MX
// blocks are the four 16-byte chunks the int8 fragment read already hands to lanes 16 g .. 16 g + 15.  The scale bytes ride
// the same LDS-DMA pipeline: with the loads of every FOURTH stage each wave issues one more global_load_lds, 16 bytes per
// lane (lane -> one of the tile's 512 operand rows, its scale bytes of four stages), into a two-slot panel; it lands under the
// same counted vmcnt waits as that stage's data, and the LOAD phase reads one dword per row tile next to the fragments.
// (One 4-byte load per wave and stage -- 64 cache lines per instruction, as many as the four data pieces together --
// held the kernel at the 128 x 128 kernel's speed.)
typedef GemmCfg<256, 256, 2, 4, true, true, OP_F8, 0, 0, EPI_PLAIN> Mx256Cfg;
constexpr int MX256_SK = 64, MX256_NST = 4, MX256_OPER = 256 * MX256_SK, MX256_PNL = 512 * 16;
constexpr int MX256_LDS = 2 * MX256_NST * MX256_OPER + 2 * MX256_PNL;   // 128 KiB + two 8 KiB scale panels

__global__ void __launch_bounds__(512) mxfp4_256_kernel(const MxParams mp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmParams &p = mp.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    gnk_gemm::tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
    char *const Abuf = smem, *const Bbuf = smem + MX256_NST * MX256_OPER, *const Sbuf = smem + 2 * MX256_NST * MX256_OPER;

    gnk_gemm::Stager64<true> sa, sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);
    // scale source of this lane: operand row 64 * wave + lane of the tile's 256 A rows followed by its 256 B rows
    const unsigned char *ssrc;
    {
        const int idx = 64 * wave + lane;
        int64_t g = idx < 256 ? m0 + idx : n0 + (idx - 256);
        const int64_t last = (idx < 256 ? p.M : p.N) - 1;
        g = g < last ? g : last;
        ssrc = idx < 256 ? mp.sa + g * mp.ldsa : mp.sb + g * mp.ldsb;
    }
    auto issue_stage = [&](int st) {                            // stage st -> buffer st & 3; every fourth stage brings its panel
        if ((st & 3) == 0 && !(GNK_DEBUG_BITS(p.debug) & 128)) {   // debug flag 128 (timing only, wrong results): no scale traffic
            __builtin_amdgcn_global_load_lds((const GNK_GLOBAL void *)ssrc, (GNK_LDS void *)(Sbuf + ((st >> 2) & 1) * MX256_PNL + wave * 1024), 16, 0, 0);
            ssrc += 16;
        }
        sa.issue(Abuf + (st & 3) * MX256_OPER, wave);
        sb.issue(Bbuf + (st & 3) * MX256_OPER, wave);
    };

    v4f acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    v4i fa[8], fb[4];
    int sca[8], scb[4];
    const int r = lane & 15, g = lane >> 4;
    const bool no_scale_reads = GNK_DEBUG_BITS(p.debug) & 1024;   // debug flag 1024 (timing only): unit scales instead of the LDS panel reads
    auto load_frags = [&](int st) {
        const char *As = Abuf + (st & 3) * MX256_OPER, *Bs = Bbuf + (st & 3) * MX256_OPER, *Ss = Sbuf + ((st >> 2) & 1) * MX256_PNL + 4 * (st & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fb[j] = *(const GNK_LDS v4i *)(const GNK_LDS char *)(Bs + gnk_k64_frag16_offset(wn * 64 + 16 * j, lane));
            scb[j] = no_scale_reads ? 127 : (int)*(const GNK_LDS unsigned char *)(Ss + (256 + wn * 64 + 16 * j + r) * 16 + g);   // ds_read_u8: the lane's own block, already at byte 0
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[i] = *(const GNK_LDS v4i *)(const GNK_LDS char *)(As + gnk_k64_frag16_offset(wm * 128 + 16 * i, lane));
            sca[i] = no_scale_reads ? 127 : (int)*(const GNK_LDS unsigned char *)(Ss + (wm * 128 + 16 * i + r) * 16 + g);
        }
    };

    const int ns = (int)(p.K / MX256_SK);
    const int npro = ns < MX256_NST ? ns : MX256_NST;
    for (int s = 0; s < npro; ++s) issue_stage(s);               // the panel of stages 0..3 is the oldest load
    if (npro == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (npro == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (npro == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gnk_gemm::pp2_phase_barrier();
    load_frags(0);
    if (npro == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 1 must have landed before the first LOAD phase reads it
    else if (npro == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gnk_gemm::pp2_phase_barrier();
    if (wm == 1) gnk_gemm::pp2_phase_barrier();   // stagger: group 1 runs one phase behind group 0

    for (int s = 0; s < ns; ++s) {
        // ---- MFMA phase (matrix B feeds the MFMA's A port: a lane then owns 4 consecutive n of one m)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v8i b8 = {fb[j][0], fb[j][1], fb[j][2], fb[j][3], 0, 0, 0, 0}, a8 = {fa[i][0], fa[i][1], fa[i][2], fa[i][3], 0, 0, 0, 0};
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8, a8, acc[i][j], FMT_FP4, FMT_FP4, 0, scb[j], 0, sca[i]);
            }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));   // pin the cluster to this phase
        __builtin_amdgcn_s_setprio(0);
        gnk_gemm::pp2_phase_barrier();
        // ---- LOAD phase
        if (s + 1 < ns) load_frags(s + 1);
        if (s + 4 < ns) {
            issue_stage(s + 4);
            // stages s+3, s+4 may stay in flight, s+2 has landed: 8 loads, 9 when one of the two brought a panel (s % 4 in {0, 1})
            if ((s & 3) < 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        gnk_gemm::pp2_phase_barrier();
    }
    if (wm == 0) gnk_gemm::pp2_phase_barrier();   // pay the stagger back so barrier counts match

    const int64_t mwave = m0 + wm * 128, nwave = n0 + wn * 64;
    if (mp.bias) {
        const int64_t ncol = nwave + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int64_t n = ncol + 16 * j + q;
                n = n < p.N ? n : p.N - 1;
                const float b = gnk_load_scalar_rt(mp.bias, mp.bias_dtype, n);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j][q] = __fadd_rn(acc[i][j][q], b);
            }
    }
    __syncthreads();
    v4f(*no_accf)[1] = nullptr;
    gnk_gemm::run_epilogue<Mx256Cfg>(p, acc, no_accf, smem, wave, lane, mwave, nwave);
}

// ------------------------------------------------------------------------------------------------------------------
// MXFP8, 256 x 256 tile: the 8-phase ping-pong program of gnk_gemm_pp.h (two 128-byte buffers per operand, one
// v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile and K-step) with real scales.  A K-step is 4 MX blocks; the lane that feeds
// rows r of a 16-row fragment with lane group g supplies the scale byte of block g (header comment), so every fragment read gets ONE
// ds_read_u8 beside it.  The scale bytes ride the LDS-DMA stream as in mxfp4_256_kernel: a panel is 16 bytes per operand row = four
// K-steps, lane -> one of the tile's 512 operand rows, issued by every wave in phase 1 of the step before the panel's first step
// (together with the B half of that tile) and complete at the same vmcnt(0) in phase 5; two panel slots.
constexpr int MX8_PNL = 512 * 16;
template <int FA, int FB> struct Mx8Cfg { typedef GemmCfg<256, 256, 2, 4, true, true, OP_F8, FA, FB, EPI_PLAIN> type; };
constexpr int MX8_256_LDS = Mx8Cfg<0, 0>::type::LDS_BYTES + 2 * MX8_PNL;     // 128 KiB + two 8 KiB scale panels

template <int FA, int FB>
__global__ void __launch_bounds__(512) mxfp8_256_kernel(const MxParams mp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mx8Cfg<FA, FB>::type Cfg;
    typedef gnk_gemm::PP<Cfg> K;
    const GemmParams &p = mp.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    gnk_gemm::tile_of_block(p.grid_m, p.grid_n, tm, tn);
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
    char *const Sbuf = smem + Cfg::LDS_BYTES;

    typename K::StA sa;
    typename K::StB sb;
    sa.init(p.A, p.lda, m0, p.M, wave, lane);
    sb.init(p.B, p.ldb, n0, p.N, wave, lane);
    // scale source of this lane: operand row 64 * wave + lane of the tile's 256 A rows followed by its 256 B rows
    const unsigned char *ssrc;
    {
        const int idx = 64 * wave + lane;
        int64_t g = idx < 256 ? m0 + idx : n0 + (idx - 256);
        const int64_t last = (idx < 256 ? p.M : p.N) - 1;
        g = g < last ? g : last;
        ssrc = idx < 256 ? mp.sa + g * mp.ldsa : mp.sb + g * mp.ldsb;
    }
    auto issue_panel = [&](int u) {                         // scale bytes of K-steps 4 u .. 4 u + 3 -> slot u & 1
        __builtin_amdgcn_global_load_lds((const GNK_GLOBAL void *)(ssrc + 16 * u), (GNK_LDS void *)(Sbuf + (u & 1) * MX8_PNL + wave * 1024), 16, 0, 0);
    };

    v4f acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    v8i fa[4], fb[2];
    int sca[4], scb[2];
    const int r = lane & 15, g = lane >> 4;
    // the lane's scale byte of K-step t for fragment rows `row0 + r` of the panel (A rows 0..255, B rows 256..511): already at byte 0.
    // (Reading the K-step's dword of two rows per ds_read2_b32 and shifting by 8 g -- 10 LDS instructions per wave and K-step instead
    // of 14 -- was 3 % SLOWER; with no scale reads at all the kernel gains 9 %, with no panel traffic 4 %: tools/mx8_ab.py.)
    auto scales_a = [&](int t, int qm) {
        const char *Ss = Sbuf + ((t >> 2) & 1) * MX8_PNL + 4 * (t & 3) + g;
#pragma unroll
        for (int i = 0; i < 4; ++i) sca[i] = (int)*(const GNK_LDS unsigned char *)(const GNK_LDS char *)(Ss + (wm * 128 + qm * 64 + 16 * i + r) * 16);
    };
    auto scales_b = [&](int t, int qn) {
        const char *Ss = Sbuf + ((t >> 2) & 1) * MX8_PNL + 4 * (t & 3) + g;
#pragma unroll
        for (int j = 0; j < 2; ++j) scb[j] = (int)*(const GNK_LDS unsigned char *)(const GNK_LDS char *)(Ss + (256 + wn * 64 + qn * 32 + 16 * j + r) * 16);
    };
    auto mfma_q = [&](auto QM, auto QN) {
        constexpr int qm = decltype(QM)::value, qn = decltype(QN)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)   // matrix B feeds the MFMA's first port (a lane then owns 4 consecutive n of one m): its format code is FB
                acc[qm * 4 + i][qn * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[qm * 4 + i][qn * 2 + j], FB, FA, 0, scb[j], 0, sca[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[qm * 4 + i][qn * 2 + j]));   // pin the cluster to this phase
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    const int nk = (int)(p.K / GNK_BK);
    char *const A0 = smem, *const B0 = smem + 2 * Cfg::A_BYTES;
    // prologue: panel 0 (the oldest load), tile 0 in full and the A half of tile 1, then the first quadrant's fragments
    issue_panel(0);
    sa.issue(A0, wave);
    sb.issue(B0, wave);
    if (nk > 1) sa.issue(A0 + Cfg::A_BYTES, wave);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    K::template load_a<0>(A0, fa, wm, lane);
    K::template load_b<0>(B0, fb, wn, lane);
    scales_a(0, 0);
    scales_b(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_setprio(1);
    if (wm == 1) __builtin_amdgcn_s_barrier();           // stagger: group 1 runs one phase behind group 0
    __builtin_amdgcn_sched_barrier(0);

    // The buffer index is a RUN-TIME value on purpose: unrolled by two (compile-time buffers, as gnk_gemm_pp.h does) the kernel keeps both
    // buffers' fragment and scale addresses live and spills 62 registers (4096^3: 122 us instead of 68-72).
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        char *const As = smem + buf * Cfg::A_BYTES, *const An = smem + (buf ^ 1) * Cfg::A_BYTES;
        char *const Bs = B0 + buf * Cfg::B_BYTES, *const Bn = B0 + (buf ^ 1) * Cfg::B_BYTES;
        mfma_q(I0(), I0());                              // p0
        K::end_mfma_phase();
        K::template load_b<1>(Bs, fb, wn, lane);         // p1
        scales_b(t, 1);
        if (t + 1 < nk) {
            if (((t + 1) & 3) == 0) issue_panel((t + 1) >> 2);
            sb.issue(Bn, wave);
        }
        K::end_load_phase();
        mfma_q(I0(), I1());                              // p2
        K::end_mfma_phase();
        K::template load_a<1>(As, fa, wm, lane);         // p3
        scales_a(t, 1);
        K::end_load_phase();
        mfma_q(I1(), I1());                              // p4
        K::end_mfma_phase();
        K::template load_b<0>(Bs, fb, wn, lane);         // p5
        scales_b(t, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); //    this wave's pieces of tile t+1 (and its panel) have landed
        if (t + 2 < nk) sa.issue(As, wave);
        K::end_load_phase();
        mfma_q(I1(), I0());                              // p6
        K::end_mfma_phase();
        if (t + 1 < nk) {                                // p7
            K::template load_a<0>(An, fa, wm, lane);
            K::template load_b<0>(Bn, fb, wn, lane);
            scales_a(t + 1, 0);
            scales_b(t + 1, 0);
        }
        K::end_load_phase();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();           // pay the stagger back so barrier counts match
    __builtin_amdgcn_sched_barrier(0);

    const int64_t mwave = m0 + wm * 128, nwave = n0 + wn * 64;
    if (mp.bias) {
        const int64_t ncol = nwave + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int64_t n = ncol + 16 * j + q;
                n = n < p.N ? n : p.N - 1;
                const float b = gnk_load_scalar_rt(mp.bias, mp.bias_dtype, n);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][j][q] = __fadd_rn(acc[i][j][q], b);
            }
    }
    __syncthreads();
    v4f(*no_accf)[1] = nullptr;
    gnk_gemm::run_epilogue<Cfg>(p, acc, no_accf, smem, wave, lane, mwave, nwave);
}

// any sizes / alignment (k % 32 == 0 by format): one output per thread, blocks in ascending order
__global__ void __launch_bounds__(256) mxfp4_generic_kernel(const MxParams mp) {
    const GemmParams &p = mp.g;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * p.N) return;
    const int64_t m = idx / p.N, n = idx - m * p.N;
    const unsigned char *a = (const unsigned char *)p.A + m * p.lda, *b = (const unsigned char *)p.B + n * p.ldb;
    const int64_t nblk = p.K / 16;   // 16 packed bytes per MX block
    float acc = 0.f;
    for (int64_t kb = 0; kb < nblk; ++kb) {
        float part = 0.f;
        for (int e = 0; e < 16; ++e) {
            const unsigned ca = a[kb * 16 + e], cb = b[kb * 16 + e];
            part += gnk_fp4_to_f32(ca & 15) * gnk_fp4_to_f32(cb & 15);
            part += gnk_fp4_to_f32(ca >> 4) * gnk_fp4_to_f32(cb >> 4);
        }
        const float s = gnk_e8m0_to_f32(mp.sa[m * mp.ldsa + kb]) * gnk_e8m0_to_f32(mp.sb[n * mp.ldsb + kb]);
        acc += part * s;
    }
    if (mp.bias) acc += gnk_load_scalar_rt(mp.bias, mp.bias_dtype, n);
    if (p.out_dtype == GNK_BF16) ((unsigned short *)p.C)[idx] = gnk_f32_to_bf16(acc);
    else if (p.out_dtype == GNK_F16) ((unsigned short *)p.C)[idx] = gnk_f32_to_f16(acc);
    else ((float *)p.C)[idx] = acc;
}

__device__ __forceinline__ float mx_f8_to_f32(unsigned c, int e5m2) {
    if (e5m2) return gnk_f16_to_f32((unsigned short)(c << 8));            // e5m2 is the top byte of an fp16
    const unsigned e = (c >> 3) & 15u, m = c & 7u;
    float v;
    if (e == 15u && m == 7u) v = __uint_as_float(0x7fc00000u);            // e4m3fn: the only NaN; no infinities
    else if (e == 0u) v = (float)m * 0x1p-9f;                             // subnormal: m / 8 * 2^-6
    else v = __uint_as_float(((e + 120u) << 23) | (m << 20));             // (1 + m / 8) * 2^(e - 7)
    return (c & 0x80u) ? -v : v;
}

// MXFP8, any sizes / alignment (k % 32 == 0 by format): one output per thread, blocks in ascending order
__global__ void __launch_bounds__(256) mxfp8_generic_kernel(const MxParams mp, int a_e5m2, int b_e5m2) {
    const GemmParams &p = mp.g;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * p.N) return;
    const int64_t m = idx / p.N, n = idx - m * p.N;
    const unsigned char *a = (const unsigned char *)p.A + m * p.lda, *b = (const unsigned char *)p.B + n * p.ldb;
    const int64_t nblk = p.K / 32;
    float acc = 0.f;
    for (int64_t kb = 0; kb < nblk; ++kb) {
        float part = 0.f;
        for (int e = 0; e < 32; ++e) part += mx_f8_to_f32(a[kb * 32 + e], a_e5m2) * mx_f8_to_f32(b[kb * 32 + e], b_e5m2);
        const float s = gnk_e8m0_to_f32(mp.sa[m * mp.ldsa + kb]) * gnk_e8m0_to_f32(mp.sb[n * mp.ldsb + kb]);
        acc += part * s;
    }
    if (mp.bias) acc += gnk_load_scalar_rt(mp.bias, mp.bias_dtype, n);
    if (p.out_dtype == GNK_BF16) ((unsigned short *)p.C)[idx] = gnk_f32_to_bf16(acc);
    else if (p.out_dtype == GNK_F16) ((unsigned short *)p.C)[idx] = gnk_f32_to_f16(acc);
    else ((float *)p.C)[idx] = acc;
}

inline bool al(const void *p, int a) { return ((uintptr_t)p % a) == 0; }

template <int FA, int FB> int launch_mxfp8(const MxParams &mp, bool big, hipStream_t st) {
    static std::atomic<bool> attr_set[64][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "mxfp8_mm: no current HIP device");
    if (!attr_set[dev][big].load(std::memory_order_acquire)) {
        const hipError_t e = big ? hipFuncSetAttribute((const void *)mxfp8_256_kernel<FA, FB>, hipFuncAttributeMaxDynamicSharedMemorySize, MX8_256_LDS)
                                 : hipFuncSetAttribute((const void *)mx_kernel<4, FA, FB>, hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS);
        if (e != hipSuccess) return gnk_fail(GNK_ERR_LAUNCH, "mxfp8_mm: cannot reserve %d bytes of LDS", big ? MX8_256_LDS : MX_LDS);
        attr_set[dev][big].store(true, std::memory_order_release);
    }
    if (big) hipLaunchKernelGGL((mxfp8_256_kernel<FA, FB>), dim3((unsigned)(mp.g.grid_m * mp.g.grid_n)), dim3(512), MX8_256_LDS, st, mp);
    else hipLaunchKernelGGL((mx_kernel<4, FA, FB>), dim3((unsigned)(mp.g.grid_m * mp.g.grid_n)), dim3(256), MX_LDS, st, mp);
    return gnk_check_launch(big ? "mxfp8_mm (256 x 256)" : "mxfp8_mm");
}

}  // namespace

extern "C" int gnk_mxfp4_mm(const void *A, const void *B, void *C, int out_dtype, const void *scale_A, const void *scale_B,
                            const void *bias, int bias_dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                            int64_t ldsa, int64_t ldsb, void *stream) {
    if (M < 0 || N < 0 || K < 0) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp4_mm: negative dimension");
    if (K % 32 != 0) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp4_mm: K = %lld is not a multiple of the MX block (32)", (long long)K);
    if (!gnk_is_float_dtype(out_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp4_mm: output dtype code %d is not bf16/f16/f32", out_dtype);
    if (bias && !gnk_is_float_dtype(bias_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp4_mm: bias dtype code %d is not bf16/f16/f32", bias_dtype);
    if (M == 0 || N == 0) return GNK_OK;
    if (!C || (K > 0 && (!A || !B || !scale_A || !scale_B))) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp4_mm: null pointer");
    if (lda < K / 2 || ldb < K / 2 || ldsa < K / 32 || ldsb < K / 32) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp4_mm: a leading dimension is shorter than its row");
    int rc = gnk_init();
    if (rc) return rc;
    MxParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.g.A = (const char *)A; mp.g.B = (const char *)B; mp.g.C = C; mp.g.M = M; mp.g.N = N; mp.g.K = K / 2; mp.g.lda = lda; mp.g.ldb = ldb;
    mp.g.out_dtype = out_dtype;
    mp.sa = (const unsigned char *)scale_A; mp.sb = (const unsigned char *)scale_B; mp.ldsa = ldsa; mp.ldsb = ldsb;
    mp.bias = bias; mp.bias_dtype = bias_dtype;
    mp.g.debug = gnk_debug_flags_get();
    hipStream_t st = (hipStream_t)stream;
    const bool fast = K >= 256 && K % 256 == 0 && al(A, 16) && al(B, 16) && lda % 16 == 0 && ldb % 16 == 0 && al(scale_A, 8) && al(scale_B, 8) &&
                      ldsa % 8 == 0 && ldsb % 8 == 0 && M < (1LL << 31) && N < (1LL << 31) && lda * 256 < (1LL << 31) && ldb * 256 < (1LL << 31) &&
                      ((M + 127) / 128) * ((N + 127) / 128) < (1LL << 31);
    // 256 x 256 ping-pong kernel once its tiles fill most of the chip (as plan_gemm does for int8 / fp8); a scale panel is 16 bytes per row: K % 512 == 0
    const bool fast256 = K >= 512 && K % 512 == 0 && al(A, 16) && al(B, 16) && lda % 16 == 0 && ldb % 16 == 0 && al(scale_A, 16) && al(scale_B, 16) &&
                         ldsa % 16 == 0 && ldsb % 16 == 0 && M < (1LL << 31) && N < (1LL << 31) && lda * 256 < (1LL << 31) && ldb * 256 < (1LL << 31);
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const int forced = mp.g.debug & 256 ? 1 : mp.g.debug & 512 ? 2 : 0;   // debug flags (A/B timing): 256 = force the 128^2 kernel, 512 = force the 256^2 kernel
    if (fast256 && tiles256 < (1LL << 31) && forced != 1 && (forced == 2 || !fast || tiles256 * 5 >= (int64_t)gnk_num_cus() * 4)) {
        mp.g.grid_m = (int)((M + 255) / 256);
        mp.g.grid_n = (int)((N + 255) / 256);
        static bool attr256[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "mxfp4_mm: no current HIP device");
        if (!attr256[dev]) {
            if (hipFuncSetAttribute((const void *)mxfp4_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MX256_LDS) != hipSuccess)
                return gnk_fail(GNK_ERR_LAUNCH, "mxfp4_mm: cannot reserve %d bytes of LDS", MX256_LDS);
            attr256[dev] = true;
        }
        hipLaunchKernelGGL(mxfp4_256_kernel, dim3((unsigned)(mp.g.grid_m * mp.g.grid_n)), dim3(512), MX256_LDS, st, mp);
        return gnk_check_launch("mxfp4_mm (256 x 256)");
    }
    if (fast) {
        mp.g.grid_m = (int)((M + 127) / 128);
        mp.g.grid_n = (int)((N + 127) / 128);
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gnk_fail(GNK_ERR_LAUNCH, "mxfp4_mm: no current HIP device");
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute((const void *)mx_kernel<8, FMT_FP4, FMT_FP4>, hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS) != hipSuccess)
                return gnk_fail(GNK_ERR_LAUNCH, "mxfp4_mm: cannot reserve %d bytes of LDS", MX_LDS);
            attr_set[dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL((mx_kernel<8, FMT_FP4, FMT_FP4>), dim3((unsigned)(mp.g.grid_m * mp.g.grid_n)), dim3(256), MX_LDS, st, mp);
        return gnk_check_launch("mxfp4_mm");
    }
    if (M * N >= (1LL << 31) * 256) return gnk_fail(GNK_ERR_UNSUPPORTED, "mxfp4_mm: output too large for the generic kernel");
    hipLaunchKernelGGL(mxfp4_generic_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, mp);
    return gnk_check_launch("mxfp4_mm (generic)");
}

extern "C" int gnk_mxfp8_mm(const void *A, int a_dtype, const void *B, int b_dtype, void *C, int out_dtype, const void *scale_A,
                            const void *scale_B, const void *bias, int bias_dtype, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldsa, int64_t ldsb, void *stream) {
    if (!gnk_is_f8(a_dtype) || !gnk_is_f8(b_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: operands must be e4m3 / e5m2 (got dtype codes %d, %d)", a_dtype, b_dtype);
    if (M < 0 || N < 0 || K < 0) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: negative dimension");
    if (K % 32 != 0) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: K = %lld is not a multiple of the MX block (32)", (long long)K);
    if (!gnk_is_float_dtype(out_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: output dtype code %d is not bf16/f16/f32", out_dtype);
    if (bias && !gnk_is_float_dtype(bias_dtype)) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: bias dtype code %d is not bf16/f16/f32", bias_dtype);
    if (M == 0 || N == 0) return GNK_OK;
    if (!C || (K > 0 && (!A || !B || !scale_A || !scale_B))) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: null pointer");
    if (lda < K || ldb < K || ldsa < K / 32 || ldsb < K / 32) return gnk_fail(GNK_ERR_BAD_ARG, "mxfp8_mm: a leading dimension is shorter than its row");
    int rc = gnk_init();
    if (rc) return rc;
    MxParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.g.A = (const char *)A; mp.g.B = (const char *)B; mp.g.C = C; mp.g.M = M; mp.g.N = N; mp.g.K = K; mp.g.lda = lda; mp.g.ldb = ldb;
    mp.g.out_dtype = out_dtype;
    mp.sa = (const unsigned char *)scale_A; mp.sb = (const unsigned char *)scale_B; mp.ldsa = ldsa; mp.ldsb = ldsb;
    mp.bias = bias; mp.bias_dtype = bias_dtype;
    hipStream_t st = (hipStream_t)stream;
    // the panel is fetched in 8-byte pieces: whole pieces per row need K % 256 == 0
    const bool fast = K >= 256 && K % 256 == 0 && al(A, 16) && al(B, 16) && lda % 16 == 0 && ldb % 16 == 0 && al(scale_A, 8) && al(scale_B, 8) &&
                      ldsa % 8 == 0 && ldsb % 8 == 0 && M < (1LL << 31) && N < (1LL << 31) && lda * 256 < (1LL << 31) && ldb * 256 < (1LL << 31) &&
                      ((M + 127) / 128) * ((N + 127) / 128) < (1LL << 31);
    // 256 x 256 ping-pong kernel once its tiles fill most of the chip (as plan_gemm does for int8 / fp8); a scale panel is 16 bytes per row
    // and four K-steps: K % 512 == 0.  gnk_set_gemm_variant: 1 forces the 128 x 128 kernel, 0 / 4 the 256 x 256 one, 2 the generic kernel.
    const bool fast256 = K >= 512 && K % 512 == 0 && al(A, 16) && al(B, 16) && lda % 16 == 0 && ldb % 16 == 0 && al(scale_A, 16) && al(scale_B, 16) &&
                         ldsa % 16 == 0 && ldsb % 16 == 0 && M < (1LL << 31) && N < (1LL << 31) && lda * 256 < (1LL << 31) && ldb * 256 < (1LL << 31);
    const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const int forced = gnk_forced_variant_get();
    const bool want256 = forced == GNK_VARIANT_256 || forced == GNK_VARIANT_256_PP8 || (forced < 0 && (!fast || tiles256 * 5 >= (int64_t)gnk_num_cus() * 4));
    const int fa = a_dtype == GNK_E5M2, fb = b_dtype == GNK_E5M2;
    if (fast256 && tiles256 < (1LL << 31) && want256) {
        mp.g.grid_m = (int)((M + 255) / 256);
        mp.g.grid_n = (int)((N + 255) / 256);
        if (!fa) return fb ? launch_mxfp8<0, 1>(mp, true, st) : launch_mxfp8<0, 0>(mp, true, st);
        return fb ? launch_mxfp8<1, 1>(mp, true, st) : launch_mxfp8<1, 0>(mp, true, st);
    }
    if (fast && forced != GNK_VARIANT_GENERIC) {
        mp.g.grid_m = (int)((M + 127) / 128);
        mp.g.grid_n = (int)((N + 127) / 128);
        if (!fa) return fb ? launch_mxfp8<0, 1>(mp, false, st) : launch_mxfp8<0, 0>(mp, false, st);
        return fb ? launch_mxfp8<1, 1>(mp, false, st) : launch_mxfp8<1, 0>(mp, false, st);
    }
    if (M * N >= (1LL << 31) * 256) return gnk_fail(GNK_ERR_UNSUPPORTED, "mxfp8_mm: output too large for the generic kernel");
    hipLaunchKernelGGL(mxfp8_generic_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, mp, a_dtype == GNK_E5M2, b_dtype == GNK_E5M2);
    return gnk_check_launch("mxfp8_mm (generic)");
}
