// int8 MFMA GEMM kernels, tile variant 5: 256 x 256, one wave per SIMD (gnk_gemm_w4.h).
#include "gnk_gemm_dispatch.h"
GNK_DEFINE_LAUNCHER_W4(gnk_launch_i8_v5, OP_I8, 0, 0)

#ifdef GNK_W4_STAMP
// Diagnostic builds only (tools/w4_ablate.sh 16): per workgroup {s_memtime, s_memrealtime} before and after the int8 main loop.
extern "C" int gnk_debug_w4_stamps(unsigned long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gnk_w4_stamps), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -1;
}
#endif
