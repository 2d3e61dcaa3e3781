// probe_mx8.hip -- which E8M0 scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which FP8 operand bytes?
// All B data = 1.0 (e4m3 0x38) with unit scales.  A data = 1.0 only in the bytes [16 h_d, 16 h_d + 16) of the lanes of group g_d
// (lane = 16 g + r), 0 elsewhere; A's scale byte = 2^1 in the lanes of group g_s only (opsel byte), 2^0 elsewhere.
// D[0][0] = 16 if the scale of group g_s does not touch those bytes, 32 if it does.  Prints the 4 x 2 x 4 table per opsel.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mx8.hip -o tools/probe_mx8 && tools/probe_mx8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPSEL>
__global__ void k(const unsigned char *A, const unsigned char *B, const unsigned char *sa, const unsigned char *sb, float *D) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int *)(A + 32 * l))[i]; b[i] = ((const int *)(B + 32 * l))[i]; }
    const int s_a = (int)sa[l] << (8 * OPSEL), s_b = (int)sb[l] << (8 * OPSEL);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OPSEL, s_a, OPSEL, s_b);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

template <int OPSEL> void table(float *dD) {
    unsigned char *dA, *dB, *dsa, *dsb;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64);
    for (int sb = 0; sb < 2; ++sb) {
        printf("opsel %d, scale varied on the %s operand: rows = data (lane group g_d, 16-byte half h_d), columns = scaled lane group g_s; entry = D[0][0]\n", OPSEL, sb ? "SECOND" : "FIRST");
        for (int gd = 0; gd < 4; ++gd) for (int hd = 0; hd < 2; ++hd) {
            printf("  data g=%d h=%d :", gd, hd);
            for (int gs = 0; gs < 4; ++gs) {
                unsigned char hA[2048], hB[2048], hsa[64], hsb[64];
                for (int l = 0; l < 64; ++l) {
                    for (int j = 0; j < 32; ++j) {
                        const unsigned char sel = ((l >> 4) == gd && (j >> 4) == hd) ? 0x38 : 0;
                        hA[32 * l + j] = sb ? 0x38 : sel;        // the operand whose scale varies carries the selective data
                        hB[32 * l + j] = sb ? sel : 0x38;
                    }
                    hsa[l] = (!sb && (l >> 4) == gs) ? 128 : 127;
                    hsb[l] = (sb && (l >> 4) == gs) ? 128 : 127;
                }
                hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
                hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
                hipLaunchKernelGGL((k<OPSEL>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
                float D[256]; for (int i = 0; i < 256; ++i) D[i] = -1;
                hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
                printf(" %5.0f", D[0]);
            }
            printf("\n");
        }
    }
}

int main() {
    float *dD; hipMalloc(&dD, 1024);
    table<0>(dD);
    table<1>(dD);
    return 0;
}
