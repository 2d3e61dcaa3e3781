"""INT8 / FP8 matmul operators of the `gn_kernels` namespace on MI355X.

Same op names, schemas, argument checks and routing as the reference
(kernels/gn_kernels/triton_mm.py:112-143, 345-441), but the device implementations launch the
hand-written gfx950 kernels of libgnk_hip.so through its C ABI instead of autotuned Triton.
As in the reference there is a Meta implementation (shape inference for torch.compile / fake
tensors) and a device implementation registered under the "CUDA" key (PyTorch-ROCm dispatches
HIP tensors there); there is deliberately NO CPU implementation.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _cabi
from ._lib import lib, lib_ops

_FP8 = (torch.float8_e4m3fn, torch.float8_e5m2)
_FLOAT = (torch.bfloat16, torch.float16, torch.float32)

lib.define("int8_mm(Tensor A, Tensor B) -> Tensor")
lib.define("scaled_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor")
lib.define("tile_scaled_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor")


# ----------------------------------------------------------------------------- int8_mm
def int8_mm(A: Tensor, B: Tensor) -> Tensor:
    """int8 x int8 -> int32, exact (reference triton_mm.py:115-118)."""
    assert A.dtype is torch.int8 and B.dtype is torch.int8
    assert A.shape[1] == B.shape[0]
    return lib_ops.int8_mm(A, B)


@torch.library.impl(lib, "int8_mm", "Meta")
def _int8_mm_meta(A: Tensor, B: Tensor):
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=torch.int32)


@_cabi.device_impl(lib, "int8_mm")
def _int8_mm_hip(A: Tensor, B: Tensor):
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty((M, N), device=A.device, dtype=torch.int32)
    L = _cabi.load()
    with _cabi.device_of(A):
        rc = L.gnk_int8_mm(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, *A.stride(), *B.stride(), _cabi.stream_of(A))
    _cabi.check(rc, "int8_mm")
    return C


def _triton_mm(A: Tensor, B: Tensor, out_dtype: torch.dtype | None = None, acc_dtype: torch.dtype | None = None):
    """Unscaled matmul entry kept under the reference's name (triton_mm.py:131-143) for
    benchmark_mm.py: int8 -> int32 and fp8 -> out_dtype go to the MFMA kernels.  16- / 32-bit float operands (benchmark_mm.py:110, an
    unchecked timing row outside the INT8/FP8 hot path) are a plain library GEMM: torch.mm (hipBLASLt / rocBLAS) with fp32 accumulation
    -- `acc_dtype=torch.float16` is accepted and accumulates in fp32 too (CDNA4 has no fp16-accumulating MFMA), which is at least as
    accurate as the reference's fp16 accumulator; the result is cast to `out_dtype` (default: A's dtype, triton_mm.py:132)."""
    assert A.shape[1] == B.shape[0]
    if A.dtype is torch.int8 and B.dtype is torch.int8 and out_dtype in (None, torch.int32) and acc_dtype in (None, torch.int32):
        if out_dtype is None:
            raise NotImplementedError("int8 output of an int8 matmul is not part of the MI355X hot path; ask for int32")
        return lib_ops.int8_mm(A, B)
    if A.dtype in _FP8 and B.dtype in _FP8 and acc_dtype in (None, torch.float32):
        from .fp8 import _fp8_mm_any

        return _fp8_mm_any(A, B, out_dtype or torch.bfloat16)
    if A.dtype in (torch.float16, torch.bfloat16, torch.float32) and B.dtype is A.dtype and acc_dtype in (None, torch.float32, torch.float16):
        return torch.mm(A, B).to(out_dtype or A.dtype)
    raise NotImplementedError(f"_triton_mm({A.dtype}, {B.dtype}, out={out_dtype}, acc={acc_dtype}) is outside the INT8/FP8 hot path")


# ----------------------------------------------------------------------------- scaled_mm
def scaled_mm(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor) -> Tensor:
    """Matmul of quantized A (M, K) and B (K, N) with the scales folded into the epilogue.

    Routing and checks follow the reference (triton_mm.py:349-373):
      * scale_A (M, 1) with scale_B (1, N) or (1, 1): row x column (or row x tensor) scaling,
        out[m, n] = (f32(acc) * scale_A[m]) * scale_B[n];
      * otherwise 2-D tile scales, scale_A (M/QM, K/QK), scale_B (K/QK, N/QN).
    A and B are both int8, or any mix of float8_e4m3fn / float8_e5m2; any strides.  The result
    has the dtype of the scales.
    """
    assert (A.dtype == B.dtype == torch.int8) or (A.dtype in _FP8 and B.dtype in _FP8)
    assert scale_A.dtype == scale_B.dtype
    assert A.ndim == B.ndim == scale_A.ndim == scale_B.ndim == 2
    assert A.shape[1] == B.shape[0]

    if scale_A.shape == (A.shape[0], 1) and scale_B.shape in ((1, B.shape[1]), (1, 1)):
        assert scale_A.is_contiguous()
        assert scale_B.is_contiguous()
        return lib_ops.scaled_mm(A, B, scale_A, scale_B)

    assert scale_A.shape[1] == scale_B.shape[0]
    return lib_ops.tile_scaled_mm(A, B, scale_A, scale_B)


@torch.library.impl(lib, "scaled_mm", "Meta")
@torch.library.impl(lib, "tile_scaled_mm", "Meta")
def _scaled_mm_meta(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor):
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=scale_A.dtype)


def _check_scale_dtype(t: Tensor, op: str) -> None:
    if t.dtype not in _FLOAT:
        raise RuntimeError(f"gn_kernels::{op}: scale dtype {t.dtype} is not bf16 / fp16 / fp32")


def _scaled_mm_launch(A: Tensor, B: Tensor, row_scale: Tensor, col_scale: Tensor, out_dtype: torch.dtype, order: int, op: str):
    M, K = A.shape
    N = B.shape[1]
    _check_scale_dtype(row_scale, op)
    C = torch.empty((M, N), device=A.device, dtype=out_dtype)
    L = _cabi.load()
    code = _cabi.DTYPE_CODE
    with _cabi.device_of(A):
        # long K, small output (grad-weight shapes): offer the library a split-K workspace if it wants one
        ws_bytes = L.gnk_scaled_mm_workspace_bytes(code[A.dtype], code[B.dtype], M, N, K) if (K >= 2048 and A.dtype is torch.int8) else 0
        if ws_bytes > 0:
            ws = torch.empty(ws_bytes, device=A.device, dtype=torch.uint8)
            rc = L.gnk_scaled_mm_ws(A.data_ptr(), code[A.dtype], B.data_ptr(), code[B.dtype], C.data_ptr(), code[out_dtype],
                                    row_scale.data_ptr(), col_scale.data_ptr(), code[row_scale.dtype], int(col_scale.numel() == 1),
                                    order, M, N, K, *A.stride(), *B.stride(), ws.data_ptr(), ws_bytes, _cabi.stream_of(A))
        else:
            rc = L.gnk_scaled_mm(A.data_ptr(), code[A.dtype], B.data_ptr(), code[B.dtype], C.data_ptr(), code[out_dtype],
                                 row_scale.data_ptr(), col_scale.data_ptr(), code[row_scale.dtype], int(col_scale.numel() == 1),
                                 order, M, N, K, *A.stride(), *B.stride(), _cabi.stream_of(A))
    _cabi.check(rc, op)
    return C


@_cabi.device_impl(lib, "scaled_mm")
def _scaled_mm_hip(A: Tensor, B: Tensor, row_scale: Tensor, col_scale: Tensor):
    # (f32(acc) * row) * col, output dtype = scale dtype (reference triton_mm.py:382-407)
    return _scaled_mm_launch(A, B, row_scale, col_scale, row_scale.dtype, 0, "scaled_mm")


# ----------------------------------------------------------------------------- scaled_mm_t (additive op, no reference counterpart)
# The reference computes grad_weight as `_dynamic_mm(input.T, grad_output).T` (mixed_precision.py:184-186, "slightly faster"): a (K, N)
# product handed to autograd as a TRANSPOSED VIEW, which AccumulateGrad then copies into the weight's layout with a strided elementwise
# kernel (5.7 ms of the Llama-1B step on this stack, 2.6 ms at 470m).  scaled_mm_t returns that transpose as a CONTIGUOUS (N, M) tensor:
# the same GEMM with the operand roles swapped (B.T @ A.T) and the epilogue's multiply order swapped with them, so every element is
# still (f32(acc) * scale_A[m]) * scale_B[n] -- bit-identical to scaled_mm(A, B, scale_A, scale_B).T.contiguous().
lib.define("scaled_mm_t(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor")


def scaled_mm_t(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor) -> Tensor:
    """== scaled_mm(A, B, scale_A, scale_B).T.contiguous() for row x column scales (scale_A (M, 1), scale_B (1, N)), without the copy."""
    assert (A.dtype == B.dtype == torch.int8) or (A.dtype in _FP8 and B.dtype in _FP8)
    assert scale_A.dtype == scale_B.dtype
    assert A.ndim == B.ndim == scale_A.ndim == scale_B.ndim == 2
    assert A.shape[1] == B.shape[0]
    assert scale_A.shape == (A.shape[0], 1) and scale_B.shape == (1, B.shape[1])
    assert scale_A.is_contiguous() and scale_B.is_contiguous()
    return lib_ops.scaled_mm_t(A, B, scale_A, scale_B)


@torch.library.impl(lib, "scaled_mm_t", "Meta")
def _scaled_mm_t_meta(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor):
    return torch.empty((B.shape[1], A.shape[0]), device=A.device, dtype=scale_A.dtype)


@_cabi.device_impl(lib, "scaled_mm_t")
def _scaled_mm_t_hip(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor):
    # C.T[n, m] = (f32(acc) * scale_A[m]) * scale_B[n]: rows are n now, so scale_A is the COLUMN scale and order 1 multiplies it first
    return _scaled_mm_launch(B.T, A.T, scale_B, scale_A, scale_A.dtype, 1, "scaled_mm_t")


@_cabi.device_impl(lib, "tile_scaled_mm")
def _tile_scaled_mm_hip(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor):
    # block sizes from integer shape ratios (reference triton_mm.py:419,434-436)
    M, K = A.shape
    N = B.shape[1]
    _check_scale_dtype(scale_A, "tile_scaled_mm")
    if scale_A.shape[0] == 0 or scale_A.shape[1] == 0 or scale_B.shape[1] == 0:
        raise RuntimeError("gn_kernels::tile_scaled_mm: empty scale tensor")
    QM, QK, QN = M // scale_A.shape[0], K // scale_A.shape[1], N // scale_B.shape[1]
    # The reference derives the block sizes by floor division and never checks the remainder (triton_mm.py:419,434-436): with M = 100
    # and 3 scale rows, row 99 would read scale row 3.  The C ABI cannot see the scale extents, so ragged tilings are refused here.
    if QM < 1 or QN < 1 or QK < 1 or M % scale_A.shape[0] or N % scale_B.shape[1] or K % scale_A.shape[1]:
        raise RuntimeError(f"gn_kernels::tile_scaled_mm: scale shapes {tuple(scale_A.shape)}, {tuple(scale_B.shape)} "
                           f"do not tile a ({M}, {K}) x ({K}, {N}) product evenly")
    C = torch.empty((M, N), device=A.device, dtype=scale_A.dtype)
    L = _cabi.load()
    code = _cabi.DTYPE_CODE
    with _cabi.device_of(A):
        rc = L.gnk_tile_scaled_mm(A.data_ptr(), code[A.dtype], B.data_ptr(), code[B.dtype], C.data_ptr(), code[C.dtype],
                                  scale_A.data_ptr(), scale_B.data_ptr(), code[scale_A.dtype], M, N, K,
                                  *A.stride(), *B.stride(), *scale_A.stride(), *scale_B.stride(), QM, QN, QK,
                                  _cabi.stream_of(A))
    _cabi.check(rc, "tile_scaled_mm")
    return C


_TILE_RESCALE_MODES = {"auto": 0, "exact": 1, "fused": 2}


def set_tile_rescale(mode: str) -> str:
    """Rescale form of tile_scaled_mm on the calling thread (include/gnk.h: gnk_set_tile_rescale); returns the previous mode.
    "auto" (default): fp32 outputs accumulate as the reference's interpreter does, acc += (f32(part) * sA) * sB with three roundings
    (triton_mm.py:327-331), bf16 / fp16 outputs as acc = fma(f32(part), sA * sB, acc) -- inside the reference's tolerance
    (benchmark_mm.py:119-121), ~1.4x faster; "exact" / "fused" force one form for every output dtype."""
    old = _cabi.load().gnk_set_tile_rescale(_TILE_RESCALE_MODES[mode])
    return {v: k for k, v in _TILE_RESCALE_MODES.items()}[old]
