"""CPU oracle for the scaled low-precision matmul path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``quantized-training_amd/``) never does.

Two restatements live here and are cross-checked against each other and against the golden
vectors generated from the reference (``tools/gen_golden.py`` -> ``tests/golden``):

* ``gnk_oracle.c``  -- plain scalar C (built by ``oracle/Makefile``), wrapped below with ctypes;
* ``torch_ref.py``  -- the reference's own CPU fallback formula in ATen ops
  (``benchmark_mm.py:37-50``), which is also what ``bench.py`` times as ``cpu_baseline``.

Parity status: PINNED against reference-generated goldens (see tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libgnk_oracle.so"

BF16, F16, F32, I8, E4M3, E5M2, I32 = range(7)

_TORCH2CODE = {
    torch.bfloat16: BF16,
    torch.float16: F16,
    torch.float32: F32,
    torch.int8: I8,
    torch.float8_e4m3fn: E4M3,
    torch.float8_e5m2: E5M2,
    torch.int32: I32,
}


def build(force: bool = False) -> Path:
    """Compile gnk_oracle.c -> libgnk_oracle.so (gcc; a second or two)."""
    src = _HERE / "gnk_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-s", "-C", str(_HERE), "libgnk_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = ctypes.CDLL(str(_LIB_PATH))
        i64, vp, ci, f32, u64 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64
        L.orc_quantize_int8.argtypes = [vp, ci, i64, i64, i64, i64, vp, i64, i64, vp, f32, ci, u64, u64, vp]
        L.orc_int8_mm.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, i64]
        L.orc_scaled_mm.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci] + [i64] * 7
        L.orc_tile_scaled_mm.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, ci] + [i64] * 14
        L.orc_fp8_mm.argtypes = [vp, ci, vp, ci, vp, ci] + [i64] * 7
        L.orc_sr_uniform.argtypes = [u64, u64, u64]
        L.orc_sr_uniform.restype = f32
        for name, arg, res in [
            ("orc_bf16_to_f32", ctypes.c_uint16, f32),
            ("orc_f32_to_bf16", f32, ctypes.c_uint16),
            ("orc_f16_to_f32", ctypes.c_uint16, f32),
            ("orc_f32_to_f16", f32, ctypes.c_uint16),
            ("orc_e4m3_to_f32", ctypes.c_uint8, f32),
            ("orc_e5m2_to_f32", ctypes.c_uint8, f32),
        ]:
            getattr(L, name).argtypes = [arg]
            getattr(L, name).restype = res
        _lib = L
    return _lib


def _cpu(t: torch.Tensor) -> torch.Tensor:
    assert t.device.type == "cpu", "the oracle only takes CPU tensors"
    return t


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def quantize_int8(x: torch.Tensor, stochastic_rounding: bool = False, *, dim: int = -1, eps: float = 1e-12,
                  seed: int = 0, offset: int = 0, noise: torch.Tensor | None = None):
    """Restates subclasses/int8.py:12-26 for a 2-D tensor of any strides.

    Returns (int8 tensor with x's strides, scale with keepdim shape in x.dtype) like the reference.
    ``noise`` (float32, same logical shape as x) replays a given U[0,1) draw; otherwise SR uses
    the shared Philox stream keyed by (seed, offset).
    """
    _cpu(x)
    assert x.ndim == 2
    dim = dim % 2
    xv = x if dim == 1 else x.T  # logical (R, C) view reducing over C
    R, C = xv.shape
    q = torch.empty_strided(x.shape, x.stride(), dtype=torch.int8)
    qv = q if dim == 1 else q.T
    scale = torch.empty(R, dtype=x.dtype)
    if noise is not None:
        nz = (noise if dim == 1 else noise.T).contiguous().float()
        mode, nptr = 2, _ptr(nz)
    else:
        mode, nptr = (1 if stochastic_rounding else 0), None
    rc = lib().orc_quantize_int8(_ptr(xv), _TORCH2CODE[x.dtype], R, C, xv.stride(0), xv.stride(1),
                                 _ptr(qv), qv.stride(0), qv.stride(1), _ptr(scale), eps, mode, seed, offset, nptr)
    assert rc == 0
    scale = scale.view(R, 1) if dim == 1 else scale.view(1, R)
    return q, scale


def int8_mm(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    _cpu(A), _cpu(B)
    M, K = A.shape
    _, N = B.shape
    C = torch.empty(M, N, dtype=torch.int32)
    lib().orc_int8_mm(_ptr(A), _ptr(B), _ptr(C), M, N, K, *A.stride(), *B.stride())
    return C


def scaled_mm(A, B, scale_A, scale_B, *, out_dtype=None, order: int = 0) -> torch.Tensor:
    """Row-scaled or tile-scaled product with the reference's routing rule (triton_mm.py:349-373)."""
    _cpu(A), _cpu(B)
    M, K = A.shape
    _, N = B.shape
    out_dtype = out_dtype or scale_A.dtype
    C = torch.empty(M, N, dtype=out_dtype)
    if tuple(scale_A.shape) == (M, 1) and tuple(scale_B.shape) in ((1, N), (1, 1)):
        sa, sb = scale_A.contiguous(), scale_B.contiguous()
        lib().orc_scaled_mm(_ptr(A), _TORCH2CODE[A.dtype], _ptr(B), _TORCH2CODE[B.dtype], _ptr(C),
                            _TORCH2CODE[out_dtype], _ptr(sa), _ptr(sb), _TORCH2CODE[sa.dtype],
                            int(sb.numel() == 1), order, M, N, K, *A.stride(), *B.stride())
    else:
        QM, QK = M // scale_A.shape[0], K // scale_A.shape[1]
        QN = N // scale_B.shape[1]
        lib().orc_tile_scaled_mm(_ptr(A), _TORCH2CODE[A.dtype], _ptr(B), _TORCH2CODE[B.dtype], _ptr(C),
                                 _TORCH2CODE[out_dtype], _ptr(scale_A), _ptr(scale_B), _TORCH2CODE[scale_A.dtype],
                                 M, N, K, *A.stride(), *B.stride(), *scale_A.stride(), *scale_B.stride(), QM, QN, QK)
    return C


def fp8_mm(A, B, *, out_dtype=torch.bfloat16) -> torch.Tensor:
    _cpu(A), _cpu(B)
    M, K = A.shape
    _, N = B.shape
    C = torch.empty(M, N, dtype=out_dtype)
    lib().orc_fp8_mm(_ptr(A), _TORCH2CODE[A.dtype], _ptr(B), _TORCH2CODE[B.dtype], _ptr(C), _TORCH2CODE[out_dtype],
                     M, N, K, *A.stride(), *B.stride())
    return C


def sr_uniform(seed: int, offset: int, idx: int) -> float:
    return lib().orc_sr_uniform(seed, offset, idx)
