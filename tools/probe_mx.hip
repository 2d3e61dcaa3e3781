// probe_mx.hip -- operand / scale lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (e2m1) and FP8 operands and real
// E8M0 block scales, checked against a CPU MX product with exact data.  Hypothesis H1: lane l = 16 g + r holds row r,
// k = 32 g .. 32 g + 31 (fp4: byte k/2 of its 16 bytes, even k in the low nibble; fp8: byte k of its 32 bytes) and its scale
// byte (opsel-selected byte of the scale VGPR) applies to exactly those 32 elements.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mx.hip -o tools/probe_mx && tools/probe_mx
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int FMT, int OPSEL>
__global__ void k(const unsigned char *A, const unsigned char *B, const unsigned char *sa, const unsigned char *sb, float *D) {
    // A: [16 rows][128 k] as bytes per lane-slot prepared by the host: lane l reads 32 bytes at A + 32 l (fp4 uses the first 16)
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int *)(A + 32 * l))[i]; b[i] = ((const int *)(B + 32 * l))[i]; }
    const int s_a = (int)sa[l] << (8 * OPSEL), s_b = (int)sb[l] << (8 * OPSEL);   // scale byte placed where opsel points
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT, FMT, OPSEL, s_a, OPSEL, s_b);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];   // D[row = 4 (l >> 4) + reg][col = l & 15]
}

static float fp4v(int c) { static const float t[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6}; return (c & 8) ? -t[c & 7] : t[c & 7]; }
static float fp8v(int c) {   // e4m3fn
    int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1 + m / 8.f, e - 7);
    return s ? -v : v;
}

template <int FMT, int OPSEL> static int run(const char *name) {
    // logical matrices: a[m][k] codes, b[k][n] codes, scales sA[m][4], sB[n][4]
    static int ac[16][128], bc[128][16];
    static unsigned char sA[16][4], sB[16][4];
    srand(7 + FMT + OPSEL);
    for (int m = 0; m < 16; ++m) for (int kk = 0; kk < 128; ++kk) {
        ac[m][kk] = FMT == 4 ? rand() & 15 : (rand() & 255); bc[kk][m] = FMT == 4 ? rand() & 15 : (rand() & 255);
        if (FMT == 0) { if ((ac[m][kk] & 0x7f) == 0x7f) ac[m][kk] = 0; if ((bc[kk][m] & 0x7f) == 0x7f) bc[kk][m] = 0; }   // no NaN
    }
    for (int m = 0; m < 16; ++m) for (int g = 0; g < 4; ++g) { sA[m][g] = 120 + rand() % 14; sB[m][g] = 120 + rand() % 14; }
    unsigned char hA[64 * 32] = {}, hB[64 * 32] = {}, hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
        const int r = l & 15, g = l >> 4;
        for (int j = 0; j < 32; ++j) {
            const int kk = 32 * g + j;
            if (FMT == 4) { hA[32 * l + j / 2] |= ac[r][kk] << (4 * (j & 1)); hB[32 * l + j / 2] |= bc[kk][r] << (4 * (j & 1)); }
            else { hA[32 * l + j] = ac[r][kk]; hB[32 * l + j] = bc[kk][r]; }
        }
        hsa[l] = sA[r][g]; hsb[l] = sB[r][g];
    }
    unsigned char *dA, *dB, *dsa, *dsb; float *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<FMT, OPSEL>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    float D[256];
    if (hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: HIP error\n", name); return 1; }
    int bad = 0, badT = 0; double worst = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double ref = 0;
        for (int g = 0; g < 4; ++g) {
            double blk = 0;
            for (int j = 0; j < 32; ++j) { const int kk = 32 * g + j; blk += (double)(FMT == 4 ? fp4v(ac[m][kk]) : fp8v(ac[m][kk])) * (FMT == 4 ? fp4v(bc[kk][n]) : fp8v(bc[kk][n])); }
            ref += blk * ldexp(1.0, sA[m][g] - 127) * ldexp(1.0, sB[n][g] - 127);
        }
        const double tol = 1e-5 * fabs(ref) + 1e-3;
        if (fabs(D[m * 16 + n] - ref) > tol) ++bad;
        if (fabs(D[n * 16 + m] - ref) > tol) ++badT;
        worst = fmax(worst, fabs(D[m * 16 + n] - ref));
    }
    printf("%s: D[m][n] mismatches %d / 256 (transposed reading: %d), worst |err| %.3g; D[0][0..3] = %g %g %g %g\n", name, bad, badT, worst, D[0], D[1], D[2], D[3]);
    return bad != 0;
}

int main() {
    int rc = 0;
    rc |= run<4, 0>("fp4 opsel 0");
    rc |= run<4, 2>("fp4 opsel 2");
    rc |= run<0, 0>("fp8 e4m3 opsel 0");
    rc |= run<0, 3>("fp8 e4m3 opsel 3");
    printf(rc ? "H1 REJECTED for at least one case\n" : "H1 holds: lane 16 g + r <-> row r, k block g; scale byte per lane\n");
    return 0;
}
