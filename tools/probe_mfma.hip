// probe_mfma.hip -- cycles per MFMA instruction on gfx950 (one workgroup of 4 or 8 waves on one CU,
// independent accumulators, operands in registers).  Build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma probe_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND> __global__ void __launch_bounds__(512) k(const int *in, int *out, long long *cyc, int iters) {
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = in[(lane * 8 + i) & 1023]; b[i] = in[(lane * 8 + i + 512) & 1023]; }
    v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    v16i ci[4] = {}; v4i di[8] = {}; v16f cf[4] = {}; v4f df[8] = {};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { ci[j & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, ci[j & 3], 0, 0, 0); }
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { di[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, di[j], 0, 0, 0); }
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { cf[j & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cf[j & 3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); }
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { df[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, df[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); }
        } else if constexpr (KIND == 4) {   // FP4 operands (cbsz = blgp = 4), unit scales
#pragma unroll
            for (int j = 0; j < 8; ++j) { df[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, df[j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); }
        } else if constexpr (KIND == 5) {   // FP4 operands, per-lane scale bytes
            const int sa = 120 + (lane & 7), sb = 124 + ((lane >> 3) & 7);
#pragma unroll
            for (int j = 0; j < 8; ++j) { df[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, df[j], 4, 4, 0, sa, 0, sb); }
        } else if constexpr (KIND == 7) {   // FP4 operands, DISTINCT operand registers per MFMA (2 a-fragments x 4 b-fragments), per-lane scales
            const int sa = 120 + (lane & 7), sb = 124 + ((lane >> 3) & 7);
            v8i a2 = {a[4], a[5], a[6], a[7], 0, 0, 0, 0}, b2 = {b[4], b[5], b[6], b[7], 0, 0, 0, 0};
            v8i b3 = {b[1], b[2], b[3], b[4], 0, 0, 0, 0}, b4v = {b[5], b[6], b[7], b[0], 0, 0, 0, 0};
            df[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, df[0], 4, 4, 0, sa, 0, sb);
            df[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b2, df[1], 4, 4, 0, sa, 0, sb + 1);
            df[2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b3, df[2], 4, 4, 0, sa, 0, sb + 2);
            df[3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b4v, df[3], 4, 4, 0, sa, 0, sb + 3);
            df[4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a2, b, df[4], 4, 4, 0, sa + 1, 0, sb);
            df[5] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a2, b2, df[5], 4, 4, 0, sa + 1, 0, sb + 1);
            df[6] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a2, b3, df[6], 4, 4, 0, sa + 1, 0, sb + 2);
            df[7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a2, b4v, df[7], 4, 4, 0, sa + 1, 0, sb + 3);
        } else if constexpr (KIND == 8) {   // int8 16x16x64 with the same distinct-operand pattern
            v4i a2 = {a[4], a[5], a[6], a[7]}, b2 = {b[4], b[5], b[6], b[7]}, b3 = {b[1], b[2], b[3], b[4]}, b4v = {b[5], b[6], b[7], b[0]};
            di[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, di[0], 0, 0, 0);
            di[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b2, di[1], 0, 0, 0);
            di[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b3, di[2], 0, 0, 0);
            di[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4v, di[3], 0, 0, 0);
            di[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b4, di[4], 0, 0, 0);
            di[5] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, di[5], 0, 0, 0);
            di[6] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b3, di[6], 0, 0, 0);
            di[7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b4v, di[7], 0, 0, 0);
        } else {                            // FP4 operands, 32 x 32 x 64
#pragma unroll
            for (int j = 0; j < 8; ++j) { cf[j & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cf[j & 3], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += ci[j][r] + (int)cf[j][r];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += di[j][r] + (int)df[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int *in, *out; long long *cyc;
    CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, 4 * 512 * 256)); CHECK(hipMalloc(&cyc, 8 * 256));
    int h[1024];
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 1024; ++i) h[i] = zero ? 0 : (rand() * 2654435761u) ^ rand();
        CHECK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
        const char *names[9] = {"i32_32x32x32_i8", "i32_16x16x64_i8", "scale_f32_32x32x64_f8", "scale_f32_16x16x128_f8", "scale_16x16x128_fp4", "scale_16x16x128_fp4_sc", "scale_32x32x64_fp4", "fp4_16x16x128_distinct", "i8_16x16x64_distinct"};
        for (int kind = 0; kind < 9; ++kind)
            for (int waves = 4; waves <= 8; waves += 4)
                for (int blocks = 1; blocks <= 256; blocks *= 256) {
                    const int iters = 20000;
                    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                    for (int rep = 0; rep < 2; ++rep) {
                        CHECK(hipEventRecord(e0));
                        switch (kind) {
                        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        default: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
                        }
                        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                    }
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                    const double per = (double)c / (iters * 8.0);            // s_memtime ticks (100 MHz const clock?) per MFMA per wave
                    const double ns = ms * 1e6 / (iters * 8.0);
                    printf("%s %-24s waves/CU %d blocks %3d: memtime ticks/MFMA %.2f  wall ns/MFMA/wave %.2f\n", zero ? "zeros " : "random", names[kind], waves, blocks, per, ns);
                }
    }
    return 0;
}
