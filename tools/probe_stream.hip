// probe_stream.hip -- what limits a 2-byte-in / 1-byte-out streaming pass on gfx950?  Variants of a bf16 -> int8
// "convert" over a (4096*S) x 4096 matrix, one wave per 4096-element row unless noted.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_stream probe_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt4(unsigned a, unsigned b) {   // 4 bf16 (two dwords) -> 4 bytes (cheap stand-in for the quantizer)
    return ((a >> 8) & 0xff) | ((a >> 16) & 0xff00) | ((b << 8) & 0xff0000) | (b & 0xff000000);
}

// MODE 0: 8 x (16-B load, 8-B store), chunk-interleaved (the current kernel's pattern)
// MODE 1: lane owns 16 consecutive elements: 2 x 16-B loads (32 B contiguous), 1 x 16-B store
// MODE 2: as 0, nontemporal stores        MODE 3: as 1, nontemporal stores
// MODE 4: read only (result folded into one store per wave)    MODE 5: write only (8-B stores)   MODE 6: write only 16-B stores
// MODE 7: as 1 with nontemporal loads as well
template <int MODE> __global__ void __launch_bounds__(256) k(const char *__restrict__ x, char *__restrict__ q, long R) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const char *xr = x + r * 8192;
    char *qr = q + r * 4096;
    if constexpr (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5) {
        v4i raw[8];
        if constexpr (MODE != 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = *(const v4i *)(xr + (i * 64 + lane) * 16);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = (v4i){lane, i, lane, i};
        }
        if constexpr (MODE == 4) {
            int s = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += raw[i][0] ^ raw[i][1] ^ raw[i][2] ^ raw[i][3];
            if (s == 0x12345678) qr[lane] = 1;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v2i o = {(int)cvt4(raw[i][0], raw[i][1]), (int)cvt4(raw[i][2], raw[i][3])};
                v2i *dst = (v2i *)(qr + (i * 64 + lane) * 8);
                if constexpr (MODE == 2) __builtin_nontemporal_store(o, dst); else *dst = o;
            }
        }
    } else {
        v4i raw[8];
        if constexpr (MODE != 6) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i *src = (const v4i *)(xr + (i * 64 + lane) * 32);
                if constexpr (MODE == 7) { raw[2 * i] = __builtin_nontemporal_load(src); raw[2 * i + 1] = __builtin_nontemporal_load(src + 1); }
                else { raw[2 * i] = src[0]; raw[2 * i + 1] = src[1]; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) raw[i] = (v4i){lane, i, lane, i};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v4i o = {(int)cvt4(raw[2 * i][0], raw[2 * i][1]), (int)cvt4(raw[2 * i][2], raw[2 * i][3]),
                     (int)cvt4(raw[2 * i + 1][0], raw[2 * i + 1][1]), (int)cvt4(raw[2 * i + 1][2], raw[2 * i + 1][3])};
            v4i *dst = (v4i *)(qr + (i * 64 + lane) * 16);
            if constexpr (MODE == 3 || MODE == 7) __builtin_nontemporal_store(o, dst); else *dst = o;
        }
    }
}

template <int MODE> float run(const char *x, char *q, long R, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, 0, x, q, R);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<MODE>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, 0, x, q, R);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    for (long S : {1L, 4L, 16L}) {
        const long R = 4096 * S;
        char *x, *q;
        CHECK(hipMalloc(&x, R * 8192)); CHECK(hipMalloc(&q, R * 4096));
        CHECK(hipMemset(x, 0x3c, R * 8192));
        const char *names[8] = {"8B stores (current)", "16B stores, 32B/lane loads", "8B nontemporal stores", "16B nontemporal stores", "read only",
                                "write only 8B", "write only 16B", "16B nt stores + nt loads"};
        float t[8];
        t[0] = run<0>(x, q, R, 50); t[1] = run<1>(x, q, R, 50); t[2] = run<2>(x, q, R, 50); t[3] = run<3>(x, q, R, 50);
        t[4] = run<4>(x, q, R, 50); t[5] = run<5>(x, q, R, 50); t[6] = run<6>(x, q, R, 50); t[7] = run<7>(x, q, R, 50);
        for (int m = 0; m < 8; ++m) {
            const double bytes = (m == 4 ? 2.0 : m == 5 || m == 6 ? 1.0 : 3.0) * R * 4096;
            printf("R=%6ld %-28s %8.2f us  %7.2f TB/s (back-to-back launches, incl. launch gaps)\n", R, names[m], t[m], bytes / t[m] * 1e-6);
        }
        CHECK(hipFree(x)); CHECK(hipFree(q));
    }
    return 0;
}
