// probe_tilewrite.hip -- how fast can 256 workgroups (one per CU, 8 waves) write a 4096 x 4096 bf16 matrix as 256 x 256 tiles?
//   mode 0: each wave writes its 128 x 64 sub-tile, 16-B stores, 8 rows x 128 B per wave-instruction (the current GEMM epilogue)
//   mode 1: each wave writes 32 full tile rows, 16-B stores, 2 rows x 512 B per wave-instruction
//   mode 2: as 1 but tiles remapped so that a workgroup's 4 waves... (not used)
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_tilewrite probe_tilewrite.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE> __global__ void __launch_bounds__(512) k(unsigned short *C, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tm = blockIdx.x / (N / 256), tn = blockIdx.x % (N / 256);
    const long m0 = (long)tm * 256, n0 = (long)tn * 256;
    const v4i d = {lane, wave, (int)blockIdx.x, 7};
    if (MODE == 0) {
        const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            *(v4i *)(C + (m0 + wm * 128 + row) * N + n0 + wn * 64 + 8 * c) = d;
        }
    } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = wave * 32 + it * 2 + (lane >> 5), c = lane & 31;
            *(v4i *)(C + (m0 + row) * N + n0 + 8 * c) = d;
        }
    }
}

int main() {
    const int N = 4096;
    unsigned short *C;
    CHECK(hipMalloc(&C, (size_t)N * N * 2));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 50;
            for (int i = 0; i < 3; ++i) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, C, N); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, C, N); }
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, C, N); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, C, N); }
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d: %.2f us per 32 MiB tile-write kernel (back-to-back launches) = %.2f TB/s\n", mode, ms * 1e3 / iters, 33.55 / (ms * 1e3 / iters));
        }
    return 0;
}
