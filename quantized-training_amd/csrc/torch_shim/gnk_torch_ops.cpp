// gnk_torch_ops.cpp -- device ("CUDA" key = HIP on ROCm) implementations of the hot gn_kernels operators, registered from C++.
//
// The operator SCHEMAS live in Python (gn_kernels/mm.py, quantize.py: the reference's own strings, triton_mm.py:112,345-346); this
// file only replaces the per-call host path dispatcher -> Python -> ctypes -> C ABI (13-22 us per call) by
// dispatcher -> C++ -> C ABI (~3 us): the reference's CUTLASS ops are registered the same way (csrc/cutlass_sm89_fp8.cu:187-190,
// TORCH_LIBRARY_IMPL(gn_kernels, CUDA, m)).  No arithmetic happens here: every function validates, allocates the output as the
// reference does (at::empty) and calls the extern "C" entry point of libgnk_hip.so declared in include/gnk.h.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm: HIP devices carry DeviceType CUDA
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include "../../../include/gnk.h"

namespace {

int dtype_code(at::ScalarType t, const char *op) {
    switch (t) {
        case at::kBFloat16: return GNK_BF16;
        case at::kHalf: return GNK_F16;
        case at::kFloat: return GNK_F32;
        case at::kChar: return GNK_I8;
        case at::kFloat8_e4m3fn: return GNK_E4M3;
        case at::kFloat8_e5m2: return GNK_E5M2;
        case at::kInt: return GNK_I32;
        default: TORCH_CHECK(false, "gn_kernels::", op, ": unsupported dtype ", t);
    }
}

bool is_float(at::ScalarType t) { return t == at::kBFloat16 || t == at::kHalf || t == at::kFloat; }

void check(int rc, const char *op) { TORCH_CHECK(rc == 0, "gn_kernels::", op, " failed (status ", rc, "): ", gnk_last_error()); }

void *stream_of(const at::Tensor &t) { return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream(); }

// ---------------------------------------------------------------------------------------------- GEMMs
at::Tensor int8_mm(const at::Tensor &A, const at::Tensor &B) {
    const int64_t M = A.size(0), K = A.size(1), N = B.size(1);
    at::Tensor C = at::empty({M, N}, A.options().dtype(at::kInt));
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(A.device());
    check(gnk_int8_mm((const int8_t *)A.data_ptr(), (const int8_t *)B.data_ptr(), (int32_t *)C.data_ptr(), M, N, K, A.stride(0), A.stride(1),
                      B.stride(0), B.stride(1), stream_of(A)),
          "int8_mm");
    return C;
}

at::Tensor scaled_mm_impl(const at::Tensor &A, const at::Tensor &B, const at::Tensor &rs, const at::Tensor &cs, at::ScalarType out, int order,
                          const char *op) {
    const int64_t M = A.size(0), K = A.size(1), N = B.size(1);
    TORCH_CHECK(is_float(rs.scalar_type()), "gn_kernels::", op, ": scale dtype ", rs.scalar_type(), " is not bf16 / fp16 / fp32");
    at::Tensor C = at::empty({M, N}, A.options().dtype(out));
    const int ad = dtype_code(A.scalar_type(), op), bd = dtype_code(B.scalar_type(), op);
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(A.device());
    // long K, small output (grad-weight shapes): offer the library a split-K workspace if it wants one
    const int64_t ws_bytes = (K >= 2048 && A.scalar_type() == at::kChar) ? gnk_scaled_mm_workspace_bytes(ad, bd, M, N, K) : 0;
    int rc;
    if (ws_bytes > 0) {
        at::Tensor ws = at::empty({ws_bytes}, A.options().dtype(at::kByte));
        rc = gnk_scaled_mm_ws(A.data_ptr(), ad, B.data_ptr(), bd, C.data_ptr(), dtype_code(out, op), rs.data_ptr(), cs.data_ptr(),
                              dtype_code(rs.scalar_type(), op), cs.numel() == 1, order, M, N, K, A.stride(0), A.stride(1), B.stride(0), B.stride(1),
                              ws.data_ptr(), ws_bytes, stream_of(A));
    } else {
        rc = gnk_scaled_mm(A.data_ptr(), ad, B.data_ptr(), bd, C.data_ptr(), dtype_code(out, op), rs.data_ptr(), cs.data_ptr(),
                           dtype_code(rs.scalar_type(), op), cs.numel() == 1, order, M, N, K, A.stride(0), A.stride(1), B.stride(0), B.stride(1),
                           stream_of(A));
    }
    check(rc, op);
    return C;
}

// (f32(acc) * row) * col, output dtype = scale dtype (reference triton_mm.py:382-407)
at::Tensor scaled_mm(const at::Tensor &A, const at::Tensor &B, const at::Tensor &rs, const at::Tensor &cs) {
    return scaled_mm_impl(A, B, rs, cs, rs.scalar_type(), 0, "scaled_mm");
}

// == scaled_mm(A, B, sA, sB).T as a contiguous tensor (gn_kernels/mm.py:scaled_mm_t): operand roles and multiply order swapped together
at::Tensor scaled_mm_t(const at::Tensor &A, const at::Tensor &B, const at::Tensor &sA, const at::Tensor &sB) {
    return scaled_mm_impl(B.t(), A.t(), sB, sA, sA.scalar_type(), 1, "scaled_mm_t");
}

// block sizes from integer shape ratios (reference triton_mm.py:419,434-436); ragged tilings are refused (see gn_kernels/mm.py)
at::Tensor tile_scaled_mm(const at::Tensor &A, const at::Tensor &B, const at::Tensor &sA, const at::Tensor &sB) {
    const int64_t M = A.size(0), K = A.size(1), N = B.size(1);
    TORCH_CHECK(is_float(sA.scalar_type()), "gn_kernels::tile_scaled_mm: scale dtype ", sA.scalar_type(), " is not bf16 / fp16 / fp32");
    TORCH_CHECK(sA.size(0) > 0 && sA.size(1) > 0 && sB.size(1) > 0, "gn_kernels::tile_scaled_mm: empty scale tensor");
    const int64_t QM = M / sA.size(0), QK = K / sA.size(1), QN = N / sB.size(1);
    TORCH_CHECK(QM >= 1 && QN >= 1 && QK >= 1 && M % sA.size(0) == 0 && N % sB.size(1) == 0 && K % sA.size(1) == 0,
                "gn_kernels::tile_scaled_mm: scale shapes ", sA.sizes(), ", ", sB.sizes(), " do not tile a (", M, ", ", K, ") x (", K, ", ", N,
                ") product evenly");
    at::Tensor C = at::empty({M, N}, A.options().dtype(sA.scalar_type()));
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(A.device());
    check(gnk_tile_scaled_mm(A.data_ptr(), dtype_code(A.scalar_type(), "tile_scaled_mm"), B.data_ptr(), dtype_code(B.scalar_type(), "tile_scaled_mm"),
                             C.data_ptr(), dtype_code(sA.scalar_type(), "tile_scaled_mm"), sA.data_ptr(), sB.data_ptr(),
                             dtype_code(sA.scalar_type(), "tile_scaled_mm"), M, N, K, A.stride(0), A.stride(1), B.stride(0), B.stride(1), sA.stride(0),
                             sA.stride(1), sB.stride(0), sB.stride(1), QM, QN, QK, stream_of(A)),
          "tile_scaled_mm");
    return C;
}

// ---------------------------------------------------------------------------------------------- quantize
bool dense2d(const at::Tensor &x) { return x.is_contiguous() || x.t().is_contiguous(); }

void sr_args(bool sr, const c10::optional<at::Tensor> &rng, uint64_t &seed) {
    seed = 0;
    if (!sr) return;
    // the Python wrappers always pass the call's Philox key (a 0-dim int64 HOST tensor drawn from torch's CPU generator); an op called
    // directly without one draws here, the same way
    const at::Tensor r = rng.has_value() ? *rng : at::randint(0, (int64_t)1 << 62, {}, at::TensorOptions().dtype(at::kLong));
    seed = (uint64_t)r.item<int64_t>();
}

std::tuple<at::Tensor, at::Tensor> quantize_int8(const at::Tensor &x, int64_t dim, bool sr, double eps, const c10::optional<at::Tensor> &rng) {
    const int64_t nd = x.dim();
    TORCH_CHECK(is_float(x.scalar_type()) && nd >= 1, "gn_kernels::quantize_int8: unsupported input (dtype ", x.scalar_type(), ", ", nd, " dims)");
    const int64_t d = ((dim % nd) + nd) % nd;
    TORCH_CHECK(nd == 2 || (d == nd - 1 && x.is_contiguous()), "gn_kernels::quantize_int8: unsupported input (shape ", x.sizes(), ", strides ",
                x.strides(), ", dim ", dim, ")");
    int64_t R, C, sxr, sxc, sqr, sqc;
    at::Tensor q;
    if (nd == 2) {
        q = dense2d(x) ? at::empty_strided(x.sizes(), x.strides(), x.options().dtype(at::kChar)) : at::empty_like(x, x.options().dtype(at::kChar));
        const int64_t o = 1 - d;
        R = x.size(o); C = x.size(d); sxr = x.stride(o); sxc = x.stride(d); sqr = q.stride(o); sqc = q.stride(d);
    } else {
        C = x.size(nd - 1);
        R = C > 0 ? x.numel() / C : 0;
        sxr = C; sxc = 1; sqr = C; sqc = 1;
        q = at::empty(x.sizes(), x.options().dtype(at::kChar));
    }
    auto sshape = x.sizes().vec();
    sshape[d] = 1;
    at::Tensor scale = at::empty(sshape, x.options());
    if (x.numel() == 0) return {q, scale.zero_()};
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t ws_bytes = gnk_quantize_int8_workspace_bytes(R, C, sxr, sxc);
    at::Tensor ws;
    if (ws_bytes > 0) ws = at::empty({ws_bytes}, x.options().dtype(at::kByte));
    uint64_t seed;
    sr_args(sr, rng, seed);
    check(gnk_quantize_int8(x.data_ptr(), dtype_code(x.scalar_type(), "quantize_int8"), R, C, sxr, sxc, (int8_t *)q.data_ptr(), sqr, sqc,
                            scale.data_ptr(), (float)eps, sr, seed, 0, ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, stream_of(x)),
          "quantize_int8");
    return {q, scale};
}

bool rows16(const at::Tensor &x) {   // shape / stride part of supports_rowwise_colmax / supports_colwise_given (gn_kernels/quantize.py)
    return x.dim() == 2 && is_float(x.scalar_type()) && x.numel() > 0 && x.stride(1) == 1 && x.stride(0) >= x.size(1) &&
           (x.stride(0) * (int64_t)x.element_size()) % 16 == 0 && x.size(1) % (16 / (int64_t)x.element_size()) == 0;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> quantize_int8_rowwise_colmax(const at::Tensor &x_in, bool sr, double eps,
                                                                           const c10::optional<at::Tensor> &rng) {
    const int code = rows16(x_in) ? dtype_code(x_in.scalar_type(), "quantize_int8_rowwise_colmax") : -1;
    TORCH_CHECK(code >= 0 && gnk_quantize_int8_rowwise_colmax_workspace_bytes(code, x_in.size(0), x_in.size(1)) > 0,
                "gn_kernels::quantize_int8_rowwise_colmax: unsupported input (shape ", x_in.sizes(), ", strides ", x_in.strides(), ", dtype ",
                x_in.scalar_type(), ")");
    const at::Tensor x = ((uintptr_t)x_in.data_ptr() % 16 == 0) ? x_in : x_in.clone(at::MemoryFormat::Contiguous);   // fresh allocations are aligned
    const int64_t R = x.size(0), C = x.size(1);
    at::Tensor q = at::empty_like(x, x.options().dtype(at::kChar));
    if (q.stride(1) != 1) q = at::empty({R, C}, x.options().dtype(at::kChar));
    at::Tensor scale = at::empty({R, 1}, x.options());
    at::Tensor colmax = at::empty({C}, x.options().dtype(at::kFloat));
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t ws_bytes = gnk_quantize_int8_rowwise_colmax_workspace_bytes(code, R, C);
    at::Tensor ws = at::empty({ws_bytes}, x.options().dtype(at::kByte));
    uint64_t seed;
    sr_args(sr, rng, seed);
    check(gnk_quantize_int8_rowwise_colmax(x.data_ptr(), code, R, C, x.stride(0), (int8_t *)q.data_ptr(), q.stride(0), scale.data_ptr(),
                                           (float *)colmax.data_ptr(), (float)eps, sr, seed, 0, ws.data_ptr(), ws_bytes, stream_of(x)),
          "quantize_int8_rowwise_colmax");
    return {q, scale, colmax};
}

std::tuple<at::Tensor, at::Tensor> quantize_int8_colwise_given(const at::Tensor &x_in, const at::Tensor &absmax, bool sr, double eps,
                                                              const c10::optional<at::Tensor> &rng) {
    TORCH_CHECK(rows16(x_in), "gn_kernels::quantize_int8_colwise_given: unsupported input (shape ", x_in.sizes(), ", strides ", x_in.strides(),
                ", dtype ", x_in.scalar_type(), ")");
    const at::Tensor x = ((uintptr_t)x_in.data_ptr() % 16 == 0) ? x_in : x_in.clone(at::MemoryFormat::Contiguous);
    const int64_t P = x.size(0), Q = x.size(1);
    TORCH_CHECK(absmax.is_cuda() && absmax.scalar_type() == at::kFloat && absmax.numel() == Q && absmax.is_contiguous(),
                "gn_kernels::quantize_int8_colwise_given: absmax must be a contiguous fp32 device tensor with one value per column");
    at::Tensor q = at::empty_like(x, x.options().dtype(at::kChar));
    if (q.stride(1) != 1) q = at::empty({P, Q}, x.options().dtype(at::kChar));
    at::Tensor scale = at::empty({1, Q}, x.options());
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
    uint64_t seed;
    sr_args(sr, rng, seed);
    check(gnk_quantize_int8_colwise_given_absmax(x.data_ptr(), dtype_code(x.scalar_type(), "quantize_int8_colwise_given"), P, Q, x.stride(0),
                                                 (int8_t *)q.data_ptr(), q.stride(0), scale.data_ptr(), (const float *)absmax.data_ptr(), (float)eps,
                                                 sr, seed, 0, stream_of(x)),
          "quantize_int8_colwise_given");
    return {q, scale};
}

}  // namespace

TORCH_LIBRARY_IMPL(gn_kernels, CUDA, m) {
    m.impl("int8_mm", &int8_mm);
    m.impl("scaled_mm", &scaled_mm);
    m.impl("scaled_mm_t", &scaled_mm_t);
    m.impl("tile_scaled_mm", &tile_scaled_mm);
    m.impl("quantize_int8", &quantize_int8);
    m.impl("quantize_int8_rowwise_colmax", &quantize_int8_rowwise_colmax);
    m.impl("quantize_int8_colwise_given", &quantize_int8_colwise_given);
}
