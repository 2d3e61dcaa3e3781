"""OCP microscaling (MX) operators on MI355X: `quantize_mx`, `mxfp4_mm`, `dequantize_mxfp4`, `pack_block_scales_nv`
(names and signatures of the reference: kernels/gn_kernels/utils.py:21-26,83-142, cutlass_mm.py:21,80-92).

The reference runs `mxfp4_mm` on a CUTLASS sm120a kernel and `quantize_mx` as eager ATen ops.  Here both are hand-written
gfx950 kernels behind the C ABI: the GEMM feeds FP4 fragments and their E8M0 block scales straight into
v_mfma_scale_f32_16x16x128_f8f6f4 (an MX block of 32 values is exactly what one lane contributes to that instruction), and
the quantizer is one HBM-bound pass.  Host tensors take the same formulas in ATen (model preparation before `.cuda()`).

Scale layout: `mxfp4_mm` takes the scales as `quantize_mx` returns them, (rows, K / 32) row-major.  CUTLASS on Blackwell
wants them pre-shuffled by `pack_block_scales_nv` (a storage detail of that hardware); a 1-D scale tensor is taken to be
in that packed order and is un-shuffled first.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _cabi
from ._lib import lib, lib_ops

FP4 = torch.float4_e2m1fn_x2
E8M0 = torch.float8_e8m0fnu
_MX_FORMAT = {FP4: 7, torch.float8_e4m3fn: _cabi.E4M3, torch.float8_e5m2: _cabi.E5M2}
_METHOD = {"ocp": 0, "nv": 1}

DTYPE_AMAX_LUT = {FP4: 6.0, torch.float8_e4m3fn: 448.0, torch.float8_e5m2: 57_344.0}
DTYPE_POW2_AMAX_LUT = {FP4: 4.0, torch.float8_e4m3fn: 256.0, torch.float8_e5m2: 32_768.0}

lib.define("quantize_mx(Tensor x, ScalarType dtype, str compute_scale_method) -> (Tensor, Tensor)")
lib.define("mxfp4_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B, Tensor? bias) -> Tensor")
lib.define("mxfp8_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B, Tensor? bias) -> Tensor")


# ------------------------------------------------------------------------------------------------ host-side formulas
def pack_block_scales_nv(scales: Tensor) -> Tensor:
    """Blackwell scale-factor order (reference utils.py:21-26): (M, N) -> flat, 128 x 4 atoms of 32 x 16 bytes."""
    M, N = scales.shape
    assert M % 128 == 0 and N % 4 == 0
    out = scales.reshape(M // 128, 128, N // 4, 4).transpose(1, 2)
    out = out.reshape(-1, 4, 32, 4).transpose(1, 2).reshape(-1, 32, 16)
    return out.flatten()


def unpack_block_scales_nv(packed: Tensor, M: int, N: int) -> Tensor:
    """Inverse of pack_block_scales_nv."""
    assert M % 128 == 0 and N % 4 == 0 and packed.numel() == M * N
    out = packed.reshape(-1, 32, 4, 4).transpose(1, 2).reshape(M // 128, N // 4, 128, 4).transpose(1, 2)
    return out.reshape(M, N).contiguous()


def _fp32_to_fp4e2m1x2(x: Tensor) -> Tensor:
    sign = (x.view(torch.int32) >> 31) & 1
    a = x.abs()
    code = torch.where(a <= 5.0, 6, 7).int()
    for bound, strict, c in ((3.5, True, 5), (2.5, False, 4), (1.75, True, 3), (1.25, False, 2), (0.75, True, 1), (0.25, False, 0)):
        code = torch.where(a < bound if strict else a <= bound, c, code)
    f4 = (sign << 3) | code
    word = f4[..., 0::8]
    for i in range(1, 8):
        word = word | (f4[..., i::8] << (4 * i))
    return word.view(FP4)


def _quantize_mx_aten(x: Tensor, dtype: torch.dtype, method: str):
    blocks = x.float().unflatten(-1, (-1, 32))
    amax = blocks.abs().amax(dim=-1)
    if method == "ocp":
        bits = ((amax.view(torch.int32) & 0x7F80_0000).view(torch.float32) / DTYPE_POW2_AMAX_LUT[dtype]).view(torch.int32) >> 23
    else:
        sb = (amax / DTYPE_AMAX_LUT[dtype]).view(torch.int32)
        e, m = sb >> 23, sb & 0x7F_FFFF
        bits = torch.where(((e > 0) & (e < 254) & (m > 0)) | ((e == 0) & (m > 0x40_0000)), e + 1, e)
    scales = bits.to(torch.int8).view(E8M0)
    y = blocks / (bits.unsqueeze(-1) << 23).view(torch.float32).clip(1e-12)
    y = y.clip(-DTYPE_AMAX_LUT[dtype], DTYPE_AMAX_LUT[dtype])
    q = _fp32_to_fp4e2m1x2(y) if dtype is FP4 else y.to(dtype)
    return q.view(x.shape[0], -1), scales


def dequantize_mxfp4(xq: Tensor, scales: Tensor) -> Tensor:
    """(M, N / 2) packed e2m1 + (M, N / 32) E8M0 -> (M, N) fp32 (reference utils.py:110-142)."""
    assert xq.dtype is FP4 and scales.dtype is E8M0
    lut = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], device=xq.device)
    M, N = xq.shape[0], xq.shape[1] * 2
    b = xq.view(torch.uint8).to(torch.int32)
    codes = torch.stack([b & 0xF, b >> 4], dim=-1).reshape(M, N)
    s = (scales.view(torch.uint8).to(torch.int32) << 23).view(torch.float32)
    return (lut[codes].view(M, N // 32, 32) * s.reshape(M, N // 32, 1)).view(M, N)


# ------------------------------------------------------------------------------------------------ quantize_mx
def quantize_mx(x: Tensor, dtype: torch.dtype, compute_scale_method: str = "ocp"):
    """(M, N) float -> (data, E8M0 scales (M, N / 32)); data is (M, N / 2) float4_e2m1fn_x2 or (M, N) float8."""
    assert dtype in (torch.float8_e4m3fn, torch.float8_e5m2, FP4)
    if compute_scale_method not in _METHOD:
        raise ValueError(f"Unsupported {compute_scale_method=}")
    if x.is_cuda:
        return lib_ops.quantize_mx(x, dtype, compute_scale_method)
    return _quantize_mx_aten(x, dtype, compute_scale_method)


@torch.library.impl(lib, "quantize_mx", "Meta")
def _quantize_mx_meta(x: Tensor, dtype: torch.dtype, compute_scale_method: str):
    M, N = x.shape
    return x.new_empty((M, N // 2 if dtype is FP4 else N), dtype=dtype), x.new_empty((M, N // 32), dtype=E8M0)


@torch.library.impl(lib, "quantize_mx", "CUDA")
def _quantize_mx_hip(x: Tensor, dtype: torch.dtype, compute_scale_method: str):
    if x.ndim != 2 or x.dtype not in (torch.bfloat16, torch.float16, torch.float32) or x.shape[1] % 32:
        raise RuntimeError(f"gn_kernels::quantize_mx: needs a 2-D bf16/f16/f32 tensor whose rows are a multiple of 32 long (got {tuple(x.shape)}, {x.dtype})")
    if x.stride(1) != 1 or (x.stride(0) * x.element_size()) % 16 or x.data_ptr() % 16:
        x = x.contiguous()
    M, N = x.shape
    q = torch.empty((M, N // 2 if dtype is FP4 else N), dtype=torch.uint8, device=x.device)
    scales = torch.empty((M, N // 32), dtype=torch.uint8, device=x.device)
    if x.numel():
        L = _cabi.load()
        with _cabi.device_of(x):
            rc = L.gnk_quantize_mx(x.data_ptr(), _cabi.DTYPE_CODE[x.dtype], M, N, x.stride(0), q.data_ptr(), _MX_FORMAT[dtype], q.stride(0),
                                   scales.data_ptr(), scales.stride(0), _METHOD[compute_scale_method], _cabi.stream_of(x))
        _cabi.check(rc, "quantize_mx")
    return q.view(dtype), scales.view(E8M0)


# ------------------------------------------------------------------------------------------------ mxfp4_mm
def mxfp4_mm(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, bias: Tensor | None = None) -> Tensor:
    """(M, K/2) x (K/2, N) packed e2m1, B.T contiguous, E8M0 block scales -> bf16 (reference cutlass_mm.py:80-86)."""
    assert A.ndim == 2 and A.dtype is FP4 and A.is_contiguous()
    assert B.ndim == 2 and B.dtype is FP4 and B.T.is_contiguous()
    assert A.shape[1] == B.shape[0]
    assert scale_A.dtype == E8M0
    assert scale_B.dtype == E8M0
    return lib_ops.mxfp4_mm(A, B, scale_A, scale_B, bias)


@torch.library.impl(lib, "mxfp4_mm", "Meta")
def _mxfp4_mm_meta(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, bias: Tensor | None = None):
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=torch.bfloat16)


def _plain_scales(s: Tensor, rows: int, nblk: int) -> Tensor:
    if s.ndim == 1:                                   # pack_block_scales_nv order
        s = unpack_block_scales_nv(s.view(torch.uint8), rows, nblk)
    if tuple(s.shape) != (rows, nblk):
        raise RuntimeError(f"gn_kernels::mxfp4_mm: scales of shape {tuple(s.shape)}, expected ({rows}, {nblk})")
    s = s.view(torch.uint8)
    return s if s.stride(1) == 1 else s.contiguous()


@torch.library.impl(lib, "mxfp4_mm", "CUDA")
def _mxfp4_mm_hip(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, bias: Tensor | None = None):
    M, Kb = A.shape
    N = B.shape[1]
    K = 2 * Kb
    if K % 32:
        raise RuntimeError(f"gn_kernels::mxfp4_mm: K = {K} is not a multiple of the MX block (32)")
    sa, sb = _plain_scales(scale_A, M, K // 32), _plain_scales(scale_B, N, K // 32)
    if bias is not None and (bias.numel() != N or bias.dtype not in (torch.bfloat16, torch.float16, torch.float32)):
        raise RuntimeError("gn_kernels::mxfp4_mm: bias must hold one bf16/f16/f32 value per output column")
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    if M == 0 or N == 0:
        return C
    if K == 0:
        C.zero_()
        return C if bias is None else C + bias.reshape(1, N).to(C.dtype)
    L = _cabi.load()
    Bt = B.T                                          # (N, K / 2) contiguous
    b = bias.contiguous() if bias is not None else None
    with _cabi.device_of(A):
        rc = L.gnk_mxfp4_mm(A.data_ptr(), Bt.data_ptr(), C.data_ptr(), _cabi.BF16, sa.data_ptr(), sb.data_ptr(),
                            b.data_ptr() if b is not None else None, _cabi.DTYPE_CODE[b.dtype] if b is not None else 0,
                            M, N, K, A.stride(0), Bt.stride(0), sa.stride(0), sb.stride(0), _cabi.stream_of(A))
    _cabi.check(rc, "mxfp4_mm")
    return C


# ------------------------------------------------------------------------------------------------ mxfp8_mm
_F8 = (torch.float8_e4m3fn, torch.float8_e5m2)


def mxfp8_mm(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, bias: Tensor | None = None) -> Tensor:
    """(M, K) x (K, N) e4m3 / e5m2 (the two may differ), B.T contiguous, E8M0 scales (M, K / 32) and (N, K / 32) as quantize_mx returns
    them -> bf16.  Additive: the reference's quantize_mx makes MXFP8 operands (utils.py:83-107) but ships no GEMM that takes them;
    same calling convention as its mxfp4_mm (cutlass_mm.py:80-86)."""
    assert A.ndim == 2 and A.dtype in _F8 and A.is_contiguous()
    assert B.ndim == 2 and B.dtype in _F8 and B.T.is_contiguous()
    assert A.shape[1] == B.shape[0]
    assert scale_A.dtype == E8M0
    assert scale_B.dtype == E8M0
    return lib_ops.mxfp8_mm(A, B, scale_A, scale_B, bias)


@torch.library.impl(lib, "mxfp8_mm", "Meta")
def _mxfp8_mm_meta(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, bias: Tensor | None = None):
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=torch.bfloat16)


@torch.library.impl(lib, "mxfp8_mm", "CUDA")
def _mxfp8_mm_hip(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, bias: Tensor | None = None):
    M, K = A.shape
    N = B.shape[1]
    if K % 32:
        raise RuntimeError(f"gn_kernels::mxfp8_mm: K = {K} is not a multiple of the MX block (32)")
    sa, sb = _plain_scales(scale_A, M, K // 32), _plain_scales(scale_B, N, K // 32)
    if bias is not None and (bias.numel() != N or bias.dtype not in (torch.bfloat16, torch.float16, torch.float32)):
        raise RuntimeError("gn_kernels::mxfp8_mm: bias must hold one bf16/f16/f32 value per output column")
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    if M == 0 or N == 0:
        return C
    if K == 0:
        C.zero_()
        return C if bias is None else C + bias.reshape(1, N).to(C.dtype)
    L = _cabi.load()
    Bt = B.T                                          # (N, K) contiguous
    b = bias.contiguous() if bias is not None else None
    with _cabi.device_of(A):
        rc = L.gnk_mxfp8_mm(A.data_ptr(), _cabi.DTYPE_CODE[A.dtype], Bt.data_ptr(), _cabi.DTYPE_CODE[B.dtype], C.data_ptr(), _cabi.BF16,
                            sa.data_ptr(), sb.data_ptr(), b.data_ptr() if b is not None else None,
                            _cabi.DTYPE_CODE[b.dtype] if b is not None else 0, M, N, K, A.stride(0), Bt.stride(0), sa.stride(0), sb.stride(0),
                            _cabi.stream_of(A))
    _cabi.check(rc, "mxfp8_mm")
    return C
