"""`gn_kernels::quantize_int8` -- the row-wise absmax INT8 quantizer as ONE device op.

The reference has no such op: `quantize_int8` (subclasses/int8.py:12-26) is ~9 eager ATen ops that
Inductor may fuse.  On MI355X the pass is a hand-written HBM-bound kernel (csrc/gnk_quantize.hip),
exposed additively under the same namespace so the tensor subclasses can call it and torch.compile
sees an opaque, functional node.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _cabi
from ._lib import lib, lib_ops

lib.define("quantize_int8(Tensor x, int dim, bool stochastic_rounding, float eps) -> (Tensor, Tensor)")

_sr_calls = 0   # advances the Philox offset between stochastic-rounding calls


def supports(x: Tensor, dim: int) -> bool:
    """Layouts the device op takes: 2-D of any strides, or N-D contiguous reducing the last dim."""
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        return False
    if x.ndim == 2:
        return True
    return x.ndim >= 1 and dim % x.ndim == x.ndim - 1 and x.is_contiguous()


def quantize_int8(x: Tensor, stochastic_rounding: bool = False, *, dim: int = -1, eps: float = 1e-12):
    """(int8 data with x's strides, scale with keepdim shape in x.dtype) -- see subclasses/int8.py."""
    return lib_ops.quantize_int8(x, dim, stochastic_rounding, eps)


def _scale_shape(x: Tensor, dim: int):
    shape = list(x.shape)
    shape[dim % x.ndim] = 1
    return shape


@torch.library.impl(lib, "quantize_int8", "Meta")
def _quantize_int8_meta(x: Tensor, dim: int, stochastic_rounding: bool, eps: float):
    return torch.empty_like(x, dtype=torch.int8), x.new_empty(_scale_shape(x, dim))


@torch.library.impl(lib, "quantize_int8", "CUDA")
def _quantize_int8_hip(x: Tensor, dim: int, stochastic_rounding: bool, eps: float):
    global _sr_calls
    if not supports(x, dim):
        raise RuntimeError(f"gn_kernels::quantize_int8: unsupported input (shape {tuple(x.shape)}, strides {x.stride()}, "
                           f"dim {dim}, dtype {x.dtype})")
    d = dim % x.ndim
    if x.ndim == 2:
        R, C = (x.shape[0], x.shape[1]) if d == 1 else (x.shape[1], x.shape[0])
        sxr, sxc = (x.stride(0), x.stride(1)) if d == 1 else (x.stride(1), x.stride(0))
        q = torch.empty_strided(x.shape, x.stride(), dtype=torch.int8, device=x.device) if _dense(x) else torch.empty_like(
            x, dtype=torch.int8)
        sqr, sqc = (q.stride(0), q.stride(1)) if d == 1 else (q.stride(1), q.stride(0))
    else:
        C = x.shape[-1]
        R = x.numel() // max(C, 1)
        sxr, sxc = C, 1
        q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
        sqr, sqc = C, 1
    scale = torch.empty(_scale_shape(x, dim), dtype=x.dtype, device=x.device)
    if x.numel() == 0:
        return q, scale.zero_()
    L = _cabi.load()
    ws_bytes = L.gnk_quantize_int8_workspace_bytes(R, C, sxr, sxc)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    seed = offset = 0
    if stochastic_rounding:
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        offset = (_sr_calls << 40) & 0xFFFFFFFFFFFFFFFF
        _sr_calls += 1
    with _cabi.device_of(x):
        rc = L.gnk_quantize_int8(x.data_ptr(), _cabi.DTYPE_CODE[x.dtype], R, C, sxr, sxc, q.data_ptr(), sqr, sqc,
                                 scale.data_ptr(), eps, int(stochastic_rounding), seed, offset,
                                 ws.data_ptr() if ws is not None else None, ws_bytes, _cabi.stream_of(x))
    _cabi.check(rc, "quantize_int8")
    return q, scale


def _dense(x: Tensor) -> bool:
    """Non-overlapping and dense in some dim order (a contiguous tensor or a transpose of one)."""
    return x.is_contiguous() or x.T.is_contiguous()
