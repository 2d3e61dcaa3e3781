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
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { df[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, df[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); }
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
        const char *names[4] = {"i32_32x32x32_i8", "i32_16x16x64_i8", "scale_f32_32x32x64_f8", "scale_f32_16x16x128_f8"};
        for (int kind = 0; kind < 4; ++kind)
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
                        default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc, iters); break;
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
