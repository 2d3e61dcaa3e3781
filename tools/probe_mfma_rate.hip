// probe_mfma_rate.hip -- the rate a bare, register-resident MFMA loop sustains on THIS chip with every CU busy and random operands
// (the clock the chip holds under matrix load is well below its 2.4 GHz maximum: MI355X_MICROARCH.md, "DVFS give-back").  bench.py runs it
// and reports the GEMM kernel's rate against this measured ceiling beside the nominal 5033 TOP/s.  One JSON line on stdout.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma_rate tools/probe_mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s\"}\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND> __global__ void __launch_bounds__(512) k(const int *in, int *out, int iters) {
    const int lane = threadIdx.x;
    v8i a, b;
    in += KIND * 1024;                              // second half: the same random bytes with the e4m3 NaN codes removed
    for (int i = 0; i < 8; ++i) { a[i] = in[(lane * 8 + i) & 1023]; b[i] = in[(lane * 8 + i + 512) & 1023]; }
    const v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    v4i di[8] = {}; v4f df[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (KIND == 0) di[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, di[j], 0, 0, 0);
            else df[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, df[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    int s = 0;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += di[j][r] + (int)df[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> int run(const int *in, int *out, int cus, double ops_per_mfma, double *tops) {
    const int iters = 40000;                       // ~60-120 ms per launch: long enough for the clock to settle
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<KIND>), dim3(cus), dim3(512), 0, 0, in, out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    *tops = (double)cus * 8 /*waves*/ * iters * 8 /*mfma per iter*/ * ops_per_mfma / (ms * 1e-3) * 1e-12;
    return 0;
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int *in, *out;
    CHECK(hipMalloc(&in, 8192)); CHECK(hipMalloc(&out, 4 * 512 * cus));
    static unsigned char h[8192];
    srand(1);
    for (int i = 0; i < 4096; ++i) {
        h[i] = (unsigned char)(rand() >> 7);
        h[4096 + i] = (h[i] & 0x7f) == 0x7f ? (unsigned char)(h[i] & 0xfe) : h[i];
    }
    CHECK(hipMemcpy(in, h, 8192, hipMemcpyHostToDevice));
    double i8 = 0, f8 = 0;
    if (run<0>(in, out, cus, 2.0 * 16 * 16 * 64, &i8)) return 1;
    if (run<1>(in, out, cus, 2.0 * 16 * 16 * 128, &f8)) return 1;
    printf("{\"cus\": %d, \"i8_16x16x64_tops\": %.1f, \"f8_16x16x128_tflops\": %.1f, \"waves_per_cu\": 8, \"operands\": \"random, register-resident\"}\n", cus, i8, f8);
    return 0;
}
