// probe_overlap.hip -- do MFMA and fp32 VALU work overlap on one gfx950 SIMD?  (the tile-scaled GEMM's question: its K-block
// rescale is 4 VALU operations per accumulator element beside 2 MFMAs per 16x16 tile.)
// Per iteration: NM x v_mfma_i32_16x16x64_i8 (independent accumulators) and / or the rescale chain cvt, mul, mul, add (+ one
// integer add that keeps the input changing) on 32 values = 160 VALU instructions.
// Modes: 0 MFMA only | 1 VALU only | 2 both in the same wave | 3 two waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o probe_overlap probe_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int NM> __global__ void __launch_bounds__(512) k(const int *in, int *out, int iters, float sa, float sb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i a4, b4;
    for (int i = 0; i < 4; ++i) { a4[i] = in[(lane * 4 + i) & 1023]; b4[i] = in[(lane * 4 + i + 512) & 1023]; }
    v4i acc[8] = {};
    int iv[32];
    float accf[32];
    for (int j = 0; j < 32; ++j) { iv[j] = in[(lane + 64 * j) & 1023] >> 12; accf[j] = 0.f; }
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int j = 0; j < NM; ++j) acc[j & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, acc[j & 7], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float f = (float)iv[j];
                f = f * sa;
                f = f * sb;
                accf[j] = accf[j] + f;
                iv[j] += 3;
            }
        }
    }
    int s = 0;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    float fs = 0.f;
    for (int j = 0; j < 32; ++j) fs += accf[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)fs;
}

template <int MODE, int NM> double run(const int *in, int *out, int blocks, int waves, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, NM>), dim3(blocks), dim3(64 * waves), 0, 0, in, out, iters, 1.0009765625f, 0.99951171875f);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    return ms * 1e6 / iters;   // ns per iteration
}

int main() {
    int *in, *out;
    CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, 4 * 512 * 256));
    int h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (rand() * 2654435761u) ^ rand();
    CHECK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int blocks = 1; blocks <= 256; blocks *= 256) {
        printf("blocks %d (one per CU), ns per iteration; NM MFMAs (16x16x64 i8) and 160 VALU instructions per iteration\n", blocks);
        const double m8 = run<0, 8>(in, out, blocks, 4, iters), m16 = run<0, 16>(in, out, blocks, 4, iters), v = run<1, 8>(in, out, blocks, 4, iters);
        printf("  1 wave/SIMD: MFMA only NM=8 %.1f  NM=16 %.1f | VALU only %.1f | both in one wave NM=8 %.1f (sum %.1f)  NM=16 %.1f (sum %.1f)\n", m8, m16, v,
               run<2, 8>(in, out, blocks, 4, iters), m8 + v, run<2, 16>(in, out, blocks, 4, iters), m16 + v);
        printf("  2 waves/SIMD: MFMA waves + VALU waves NM=8 %.1f  NM=16 %.1f | both in every wave NM=8 %.1f (2 x sum %.1f)  NM=16 %.1f (2 x sum %.1f) | VALU only x2 %.1f | MFMA only x2 NM=16 %.1f\n",
               run<3, 8>(in, out, blocks, 8, iters), run<3, 16>(in, out, blocks, 8, iters), run<2, 8>(in, out, blocks, 8, iters), 2 * (m8 + v),
               run<2, 16>(in, out, blocks, 8, iters), 2 * (m16 + v), run<1, 8>(in, out, blocks, 8, iters), run<0, 16>(in, out, blocks, 8, iters));
    }
    return 0;
}
