"""ATen-level CPU restatement of the reference path -- TEST INFRASTRUCTURE ONLY (see __init__).

This is the "reference's CPU path" named by BASELINE.json: the reference ships no CPU impl of
``gn_kernels::scaled_mm`` (only Meta + CUDA, kernels/gn_kernels/triton_mm.py:376-441), so its
CPU-runnable form is the fallback formula of benchmark_mm.py:37-50 fed by the quantizer of
subclasses/int8.py:12-26.  bench.py times ``dynamic_int8_mm_cpu`` as ``cpu_baseline`` ("port").
"""
from __future__ import annotations

import torch
from torch import Tensor


def quantize_int8(x: Tensor, stochastic_rounding: bool = False, *, dim: int = -1, eps: float = 1e-12,
                  noise: Tensor | None = None):
    """subclasses/int8.py:12-26 in ATen ops (fp32 math, true divisions, round-half-even)."""
    xf = x.to(torch.float32)
    scale = xf.abs().amax(dim, keepdim=True) / 127          # :17
    xf = xf / scale.clamp_min(eps)                           # :18
    if stochastic_rounding:
        u = torch.rand_like(xf) if noise is None else noise  # :21
        xf = torch.floor(xf + u)
    else:
        xf = torch.round(xf)                                 # :23
    return xf.clamp(-128, 127).to(torch.int8), scale.to(x.dtype)   # :25-26


def _expand(scale: Tensor, rows: int, cols: int) -> Tensor:
    s = scale.to(torch.float32)
    s = s.repeat_interleave(rows // s.shape[0], 0)
    return s.repeat_interleave(cols // s.shape[1], 1)


def scaled_mm_dequant(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor) -> Tensor:
    """benchmark_mm.py:46-50 (``scaled_mm_ref``): dequantize both operands then fp32 matmul -> bf16.
    This is the checker the reference's own benchmark asserts against (tolerance-based)."""
    a = A.to(torch.float32) * _expand(scale_A, *A.shape)
    b = B.to(torch.float32) * _expand(scale_B, *B.shape)
    return (a @ b).to(torch.bfloat16)


def scaled_mm_intmm(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor) -> Tensor:
    """benchmark_mm.py:41 (``scaled_mm_inductor``, uncompiled): torch._int_mm then scale in fp32.
    Note the multiply order here is (acc * col) * row; the Triton kernel uses (acc * row) * col."""
    if A.dtype == torch.int8:
        acc = torch._int_mm(A, B)
    else:
        acc = A.to(torch.float32) @ B.to(torch.float32)
    return (acc * scale_B.to(torch.float32) * scale_A.to(torch.float32)).to(scale_A.dtype)


def scaled_mm_exact(A: Tensor, B: Tensor, scale_A: Tensor, scale_B: Tensor, out_dtype=None) -> Tensor:
    """Bit-level restatement of the Triton row-scaled epilogue (triton_mm.py:210-216):
    (f32(acc) * row) * col, cast on store.  INT8 only (exact int32 accumulator)."""
    assert A.dtype == torch.int8
    acc = torch._int_mm(A.contiguous(), B) if _int_mm_ok(A, B) else (A.to(torch.int32) @ B.to(torch.int32))
    out = (acc.to(torch.float32) * scale_A.to(torch.float32)) * scale_B.to(torch.float32)
    return out.to(out_dtype or scale_A.dtype)


def _int_mm_ok(A: Tensor, B: Tensor) -> bool:
    M, K = A.shape
    N = B.shape[1]
    return M > 16 and K % 8 == 0 and N % 8 == 0


def dynamic_int8_mm_cpu(A: Tensor, W: Tensor) -> Tensor:
    """Config 1 of BASELINE.json: row-wise dynamic INT8 quantize of both operands + scaled int8
    A @ W.T through torch._int_mm."""
    A_i8, sa = quantize_int8(A, dim=1)
    W_i8, sw = quantize_int8(W, dim=1)
    return (torch._int_mm(A_i8, W_i8.T).to(torch.float32) * sa.to(torch.float32) * sw.T.to(torch.float32)).to(
        torch.bfloat16
    )
