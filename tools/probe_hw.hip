// probe_hw.hip -- one-shot hardware semantics probe for gfx950 (diagnostic tool, not shipped):
//   1. ds_read_b64_tr_b8 gather pattern,
//   2. v_mfma_i32_16x16x64_i8 operand slot convention + C/D layout,
//   3. v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3) with unit scales and the "two 64-k halves" slot convention.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define LDSAS __attribute__((address_space(3)))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void k_tr8(unsigned* out_lo, unsigned* out_hi) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)(i & 0xff);
    __syncthreads();
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDSAS v2i*)(lds + threadIdx.x * 8));   // lane l addresses bytes 8l..8l+7 (< 512: low byte unique per 256)
    out_lo[threadIdx.x * 2] = r[0]; out_lo[threadIdx.x * 2 + 1] = r[1];
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)(i >> 8);
    __syncthreads();
    r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDSAS v2i*)(lds + threadIdx.x * 8));
    out_hi[threadIdx.x * 2] = r[0]; out_hi[threadIdx.x * 2 + 1] = r[1];
}

// A_op: 16 x 64 int8 (row-major), B_op: 64 x 16 int8 stored as Bt[16][64]; D: 16 x 16 int32
__global__ void k_mfma_i8(const signed char* A, const signed char* Bt, int* D) {
    const int l = threadIdx.x;
    v4i a = *(const v4i*)(A + (l & 15) * 64 + 16 * (l >> 4));
    v4i b = *(const v4i*)(Bt + (l & 15) * 64 + 16 * (l >> 4));
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];   // row = 4*(l>>4)+r, col = l&15
}

// fp8 e4m3: A_op 16 x 128, Bt 16 x 128 (bytes). slot (g, byte j): k = 64*(j/16) + 16*g + j%16
__global__ void k_mfma_f8(const unsigned char* A, const unsigned char* Bt, float* D, int scale_word, int fa, int fb) {
    const int l = threadIdx.x;
    const int g = l >> 4;
    v4i alo = *(const v4i*)(A + (l & 15) * 128 + 16 * g), ahi = *(const v4i*)(A + (l & 15) * 128 + 64 + 16 * g);
    v4i blo = *(const v4i*)(Bt + (l & 15) * 128 + 16 * g), bhi = *(const v4i*)(Bt + (l & 15) * 128 + 64 + 16 * g);
    v8i a = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
    v8i b = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
    v4f c = {0, 0, 0, 0};
    if (fa == 0 && fb == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_word, 0, scale_word);
    else if (fa == 0 && fb == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 1, 0, scale_word, 0, scale_word);
    else if (fa == 1 && fb == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 0, 0, scale_word, 0, scale_word);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, scale_word, 0, scale_word);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

static float e4m3(unsigned char b) { int s = b >> 7, e = (b >> 3) & 15, m = b & 7; float v = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10); return s ? -v : v; }
static float e5m2(unsigned char b) { int s = b >> 7, e = (b >> 2) & 31, m = b & 3; float v = e == 0 ? ldexpf((float)m, -16) : ldexpf((float)(4 + m), e - 17); return s ? -v : v; }

int main() {
    // ---- 1. tr8
    unsigned *lo, *hi; CK(hipMalloc(&lo, 512)); CK(hipMalloc(&hi, 512));
    hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, lo, hi); CK(hipDeviceSynchronize());
    unsigned hlo[128], hhi[128]; CK(hipMemcpy(hlo, lo, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(hhi, hi, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int b = 0; b < 8; ++b) {
        int src = ((hlo[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff) | (((hhi[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff) << 8);
        int grp = l & ~15, i = l & 15;
        int want = (grp + 2 * b + (i >> 3)) * 8 + (i & 7);
        if (src != want) { if (bad < 8) printf("tr8: lane %d byte %d reads LDS byte %d (src lane %d byte %d), expected %d\n", l, b, src, src / 8, src % 8, want); ++bad; }
    }
    printf("PROBE tr8: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    if (bad) { printf("tr8 table (lane: source byte addresses):\n"); for (int l = 0; l < 32; ++l) { printf("  %2d:", l); for (int b = 0; b < 8; ++b) printf(" %4d", ((hlo[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff) | (((hhi[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff) << 8)); printf("\n"); } }

    // ---- 2. i8 mfma
    srand(1);
    std::vector<signed char> A(16 * 64), Bt(16 * 64);
    for (auto& v : A) v = (signed char)(rand() % 255 - 127);
    for (auto& v : Bt) v = (signed char)(rand() % 255 - 127);
    signed char *dA, *dB; int* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_i8, dim3(1), dim3(64), 0, 0, dA, dB, dD); CK(hipDeviceSynchronize());
    int D[256]; CK(hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost));
    bad = 0; int badT = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        int acc = 0; for (int k = 0; k < 64; ++k) acc += (int)A[i * 64 + k] * (int)Bt[j * 64 + k];
        if (D[i * 16 + j] != acc) ++bad;
        if (D[j * 16 + i] != acc) ++badT;
    }
    printf("PROBE mfma_i32_16x16x64_i8: %s (mismatch %d, as-transposed %d)\n", bad ? "FAIL" : "PASS", bad, badT);

    // ---- 3. f8f6f4 mfma
    for (int fa = 0; fa < 2; ++fa) for (int fb = 0; fb < 2; ++fb) for (int sw = 0; sw < 2; ++sw) {
        std::vector<unsigned char> A8(16 * 128), B8(16 * 128);
        for (auto& v : A8) { do v = (unsigned char)(rand() & 0xff); while ((v & 0x7f) >= (fa ? 0x7c : 0x7f)); }
        for (auto& v : B8) { do v = (unsigned char)(rand() & 0xff); while ((v & 0x7f) >= (fb ? 0x7c : 0x7f)); }
        unsigned char *dA8, *dB8; float* dF; CK(hipMalloc(&dA8, 2048)); CK(hipMalloc(&dB8, 2048)); CK(hipMalloc(&dF, 1024));
        CK(hipMemcpy(dA8, A8.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB8, B8.data(), 2048, hipMemcpyHostToDevice));
        int word = sw ? 0x7f7f7f7f : 0;
        hipLaunchKernelGGL(k_mfma_f8, dim3(1), dim3(64), 0, 0, dA8, dB8, dF, word, fa, fb); CK(hipDeviceSynchronize());
        float F[256]; CK(hipMemcpy(F, dF, 1024, hipMemcpyDeviceToHost));
        double maxrel = 0, ratio = 0; int n = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double acc = 0, mag = 0;
            for (int k = 0; k < 128; ++k) { double p = (double)(fa ? e5m2(A8[i * 128 + k]) : e4m3(A8[i * 128 + k])) * (double)(fb ? e5m2(B8[j * 128 + k]) : e4m3(B8[j * 128 + k])); acc += p; mag += fabs(p); }
            double rel = fabs(F[i * 16 + j] - acc) / (mag + 1e-30);
            if (rel > maxrel) maxrel = rel;
            if (fabs(acc) > 1e-3 * mag) { ratio += F[i * 16 + j] / acc; ++n; }
        }
        printf("PROBE mfma_scale_f8f6f4 fmtA=%d fmtB=%d scale_word=0x%08x: max |err|/sum|prod| = %.3e, mean got/want = %.6g  -> %s\n", fa, fb, word, maxrel, n ? ratio / n : 0.0, maxrel < 1e-5 ? "PASS" : "FAIL");
    }
    return 0;
}
