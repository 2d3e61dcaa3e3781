// What a dispatch of the GEMM's shape costs with no work in it: 256 workgroups x 256 threads, with / without 130 KiB of LDS and a
// 512-register wave, event-timed per launch (as bench.py times kernels) and as the average of back-to-back launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256) k_null(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void __launch_bounds__(256) k_lds(int *p) { extern __shared__ char smem[]; if (p && threadIdx.x == 9999) *p = smem[0]; }
__global__ void __launch_bounds__(256) k_lds_regs(int *p) {
    extern __shared__ char smem[];
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
    if (p && threadIdx.x == 9999) *p = smem[0];
}
template <class F> static void run(const char *name, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int i = 0; i < 50; ++i) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f); }
    std::sort(ts.begin(), ts.end());
    hipEventRecord(a); for (int i = 0; i < 200; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s per-launch events: median %6.2f us   200 back to back: %6.2f us each\n", name, ts[ts.size() / 2], ms * 1e3f / 200);
}
int main() {
    hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 133120);
    hipFuncSetAttribute((const void *)k_lds_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 133120);
    run("null kernel, 256 x 256", [] { hipLaunchKernelGGL(k_null, dim3(256), dim3(256), 0, 0, nullptr); });
    run("null kernel, 1024 x 256", [] { hipLaunchKernelGGL(k_null, dim3(1024), dim3(256), 0, 0, nullptr); });
    run("130 KiB LDS, 256 x 256", [] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 133120, 0, nullptr); });
    run("130 KiB LDS + 512 registers, 256 x 256", [] { hipLaunchKernelGGL(k_lds_regs, dim3(256), dim3(256), 133120, 0, nullptr); });
    run("130 KiB LDS + 512 registers, 1024 x 256", [] { hipLaunchKernelGGL(k_lds_regs, dim3(1024), dim3(256), 133120, 0, nullptr); });
    return 0;
}
