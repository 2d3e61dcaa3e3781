"""FP8 matmul operators kept under the reference's CUTLASS op names
(kernels/gn_kernels/cutlass_mm.py:17-20,36-47,59-77; csrc/cutlass_sm89_fp8.cu, cutlass_sm120a_fp8.cu).

On MI355X they run on the same LDS-DMA MFMA kernel family as `scaled_mm`: the scaled MFMAs with unit block scales (OCP e4m3fn /
e5m2 are native on gfx950) -- v_mfma_scale_f32_16x16x128_f8f6f4 in the 8-phase 256 x 256 and the 128 x 128 kernels,
v_mfma_scale_f32_32x32x64_f8f6f4 in the four-stage 256 x 256 kernel (A.T@B), csrc/gnk_gemm_dispatch.h.
`cutlass_fp8_mm` / `cutlass_scaled_fp8_mm` are the sm120 spellings; they alias the same kernels.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _cabi
from ._lib import lib, lib_ops
from .mm import _scaled_mm_launch

_FP8 = (torch.float8_e4m3fn, torch.float8_e5m2)

lib.define("fp8_mm(Tensor A, Tensor B) -> Tensor")
lib.define("cutlass_fp8_mm(Tensor A, Tensor B) -> Tensor")
lib.define("scaled_fp8_mm(Tensor A, Tensor B, Tensor row_scale, Tensor col_scale) -> Tensor")
lib.define("cutlass_scaled_fp8_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor")


def fp8_mm(A: Tensor, B: Tensor) -> Tensor:
    """e4m3 x e4m3 -> bf16, A row-major and B column-major (reference cutlass_mm.py:36-42).
    The reference picks the `cutlass_` op on compute capability 12; there is no such device here."""
    assert A.ndim == 2 and A.dtype is torch.float8_e4m3fn and A.is_contiguous()
    assert B.ndim == 2 and B.dtype is torch.float8_e4m3fn and B.T.is_contiguous()
    return lib_ops.fp8_mm(A, B)


def scaled_fp8_mm(A: Tensor, B: Tensor, row_scale: Tensor, col_scale: Tensor) -> Tensor:
    """fp8 x fp8 with fp32 row (M) and column (N) scales -> bf16 (reference cutlass_mm.py:59-71)."""
    assert A.ndim == 2 and A.is_contiguous()
    assert B.ndim == 2 and B.T.is_contiguous()
    assert A.dtype == B.dtype
    assert A.dtype in _FP8
    assert row_scale.dtype == col_scale.dtype == torch.float32
    assert row_scale.squeeze().shape == (A.shape[0],)
    assert col_scale.squeeze().shape == (B.shape[1],)
    return lib_ops.scaled_fp8_mm(A, B, row_scale, col_scale)


@torch.library.impl(lib, "fp8_mm", "Meta")
@torch.library.impl(lib, "cutlass_fp8_mm", "Meta")
def _fp8_mm_meta(A: Tensor, B: Tensor) -> Tensor:
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=torch.bfloat16)


# The reference's Meta impl for scaled_fp8_mm returns row_scale.dtype (fp32) while its CUDA kernel
# returns bf16 (cutlass_sm89_fp8.cu:164); the device behaviour is the one kept here.
@torch.library.impl(lib, "scaled_fp8_mm", "Meta")
@torch.library.impl(lib, "cutlass_scaled_fp8_mm", "Meta")
def _scaled_fp8_mm_meta(A: Tensor, B: Tensor, row_scale: Tensor, col_scale: Tensor) -> Tensor:
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=torch.bfloat16)


def _fp8_mm_any(A: Tensor, B: Tensor, out_dtype: torch.dtype) -> Tensor:
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty((M, N), device=A.device, dtype=out_dtype)
    L = _cabi.load()
    code = _cabi.DTYPE_CODE
    with _cabi.device_of(A):
        rc = L.gnk_fp8_mm(A.data_ptr(), code[A.dtype], B.data_ptr(), code[B.dtype], C.data_ptr(), code[out_dtype],
                          M, N, K, *A.stride(), *B.stride(), _cabi.stream_of(A))
    _cabi.check(rc, "fp8_mm")
    return C


@torch.library.impl(lib, "fp8_mm", "CUDA")
@torch.library.impl(lib, "cutlass_fp8_mm", "CUDA")
def _fp8_mm_hip(A: Tensor, B: Tensor) -> Tensor:
    return _fp8_mm_any(A, B, torch.bfloat16)


@torch.library.impl(lib, "scaled_fp8_mm", "CUDA")
@torch.library.impl(lib, "cutlass_scaled_fp8_mm", "CUDA")
def _scaled_fp8_mm_hip(A: Tensor, B: Tensor, row_scale: Tensor, col_scale: Tensor) -> Tensor:
    # CUTLASS epilogue order: (acc * col) * row in fp32, then bf16 (cutlass_sm89_fp8.cu:100-107)
    rs = row_scale.reshape(-1).contiguous()
    cs = col_scale.reshape(-1).contiguous()
    return _scaled_mm_launch(A, B, rs, cs, torch.bfloat16, 1, "scaled_fp8_mm")
