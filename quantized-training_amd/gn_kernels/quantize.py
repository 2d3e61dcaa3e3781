"""`gn_kernels::quantize_int8` -- the row-wise absmax INT8 quantizer as ONE device op.

The reference has no such op: `quantize_int8` (subclasses/int8.py:12-26) is ~9 eager ATen ops that
Inductor may fuse.  On MI355X the pass is a hand-written HBM-bound kernel (csrc/gnk_quantize.hip),
exposed additively under the same namespace so the tensor subclasses can call it and torch.compile
sees an opaque, functional node.
"""
from __future__ import annotations

import threading

import torch
from torch import Tensor

from . import _cabi
from ._lib import lib, lib_ops

lib.define("quantize_int8(Tensor x, int dim, bool stochastic_rounding, float eps) -> (Tensor, Tensor)")
# Fused dual-direction pair: every X, W and dY of a mixed-precision Linear is quantized along BOTH dims
# per step (reference mixed_precision.py:100-108 with :154,:178,:186).  `quantize_int8_rowwise_colmax` is quantize_int8(x,
# dim=1) that also returns the fp32 absmax of every column from the same read; `quantize_int8_colwise_given` is
# quantize_int8(x, dim=0) for a caller who already holds that absmax (one read instead of two).
lib.define("quantize_int8_rowwise_colmax(Tensor x, bool stochastic_rounding, float eps) -> (Tensor, Tensor, Tensor)")
lib.define("quantize_int8_colwise_given(Tensor x, Tensor absmax, bool stochastic_rounding, float eps) -> (Tensor, Tensor)")

_sr_calls = 0   # advances the Philox offset between stochastic-rounding calls
_sr_lock = threading.Lock()


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
    seed, offset = _sr_args(stochastic_rounding)
    with _cabi.device_of(x):
        rc = L.gnk_quantize_int8(x.data_ptr(), _cabi.DTYPE_CODE[x.dtype], R, C, sxr, sxc, q.data_ptr(), sqr, sqc,
                                 scale.data_ptr(), eps, int(stochastic_rounding), seed, offset,
                                 ws.data_ptr() if ws is not None else None, ws_bytes, _cabi.stream_of(x))
    _cabi.check(rc, "quantize_int8")
    return q, scale


def _dense(x: Tensor) -> bool:
    """Non-overlapping and dense in some dim order (a contiguous tensor or a transpose of one)."""
    return x.is_contiguous() or x.T.is_contiguous()


# ------------------------------------------------------------------------------ fused dual-direction quantization
def _sr_args(stochastic_rounding: bool):
    """(seed, offset) of one stochastic-rounding call: every call gets its own 2^40-element slice of the Philox stream.  Forward runs
    on the caller's thread and backward on the autograd engine's, so the call counter is taken under a lock."""
    global _sr_calls
    if not stochastic_rounding:
        return 0, 0
    with _sr_lock:
        calls = _sr_calls
        _sr_calls += 1
    return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, (calls << 40) & 0xFFFFFFFFFFFFFFFF


def supports_rowwise_colmax(x: Tensor) -> bool:
    """2-D device tensor with contiguous, 16-byte aligned rows of at most 6144 (16-bit) / 3072 (fp32) elements."""
    if not (x.is_cuda and x.ndim == 2 and x.dtype in _cabi.DTYPE_CODE and x.dtype.is_floating_point and x.numel() > 0):
        return False
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32) or x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        return False
    esz = x.element_size()
    if x.data_ptr() % 16 or (x.stride(0) * esz) % 16:
        return False
    return _cabi.load().gnk_quantize_int8_rowwise_colmax_workspace_bytes(_cabi.DTYPE_CODE[x.dtype], x.shape[0], x.shape[1]) > 0


def supports_colwise_given(x: Tensor) -> bool:
    if not (x.is_cuda and x.ndim == 2 and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.numel() > 0):
        return False
    esz = x.element_size()
    return x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.shape[1] % (16 // esz) == 0 and x.data_ptr() % 16 == 0 and (x.stride(0) * esz) % 16 == 0


def quantize_int8_rowwise_colmax(x: Tensor, stochastic_rounding: bool = False, *, eps: float = 1e-12):
    """(int8 data, scale (R, 1) in x.dtype, column absmax (C,) fp32): quantize_int8(x, dim=1) plus amax|x| over dim 0."""
    return lib_ops.quantize_int8_rowwise_colmax(x, stochastic_rounding, eps)


def quantize_int8_colwise_given(x: Tensor, absmax: Tensor, stochastic_rounding: bool = False, *, eps: float = 1e-12):
    """quantize_int8(x, dim=0) given amax|x| over dim 0 (fp32, (C,)): (int8 data, scale (1, C) in x.dtype)."""
    return lib_ops.quantize_int8_colwise_given(x, absmax, stochastic_rounding, eps)


@torch.library.impl(lib, "quantize_int8_rowwise_colmax", "Meta")
def _rowwise_colmax_meta(x: Tensor, stochastic_rounding: bool, eps: float):
    return torch.empty_like(x, dtype=torch.int8), x.new_empty((x.shape[0], 1)), x.new_empty((x.shape[1],), dtype=torch.float32)


@torch.library.impl(lib, "quantize_int8_colwise_given", "Meta")
def _colwise_given_meta(x: Tensor, absmax: Tensor, stochastic_rounding: bool, eps: float):
    return torch.empty_like(x, dtype=torch.int8), x.new_empty((1, x.shape[1]))


@torch.library.impl(lib, "quantize_int8_rowwise_colmax", "CUDA")
def _rowwise_colmax_hip(x: Tensor, stochastic_rounding: bool, eps: float):
    if not supports_rowwise_colmax(x):
        raise RuntimeError(f"gn_kernels::quantize_int8_rowwise_colmax: unsupported input (shape {tuple(x.shape)}, strides {x.stride()}, dtype {x.dtype})")
    R, C = x.shape
    L = _cabi.load()
    code = _cabi.DTYPE_CODE[x.dtype]
    q = torch.empty_like(x, dtype=torch.int8)
    if q.stride(1) != 1:
        q = torch.empty((R, C), dtype=torch.int8, device=x.device)
    scale = torch.empty((R, 1), dtype=x.dtype, device=x.device)
    colmax = torch.empty((C,), dtype=torch.float32, device=x.device)
    ws_bytes = L.gnk_quantize_int8_rowwise_colmax_workspace_bytes(code, R, C)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    seed, offset = _sr_args(stochastic_rounding)
    with _cabi.device_of(x):
        rc = L.gnk_quantize_int8_rowwise_colmax(x.data_ptr(), code, R, C, x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(),
                                                colmax.data_ptr(), eps, int(stochastic_rounding), seed, offset, ws.data_ptr(), ws_bytes,
                                                _cabi.stream_of(x))
    _cabi.check(rc, "quantize_int8_rowwise_colmax")
    return q, scale, colmax


@torch.library.impl(lib, "quantize_int8_colwise_given", "CUDA")
def _colwise_given_hip(x: Tensor, absmax: Tensor, stochastic_rounding: bool, eps: float):
    if not supports_colwise_given(x):
        raise RuntimeError(f"gn_kernels::quantize_int8_colwise_given: unsupported input (shape {tuple(x.shape)}, strides {x.stride()}, dtype {x.dtype})")
    P, Q = x.shape
    if not (absmax.is_cuda and absmax.dtype is torch.float32 and absmax.numel() == Q and absmax.is_contiguous()):
        raise RuntimeError("gn_kernels::quantize_int8_colwise_given: absmax must be a contiguous fp32 device tensor with one value per column")
    L = _cabi.load()
    q = torch.empty_like(x, dtype=torch.int8)
    if q.stride(1) != 1:
        q = torch.empty((P, Q), dtype=torch.int8, device=x.device)
    scale = torch.empty((1, Q), dtype=x.dtype, device=x.device)
    seed, offset = _sr_args(stochastic_rounding)
    with _cabi.device_of(x):
        rc = L.gnk_quantize_int8_colwise_given_absmax(x.data_ptr(), _cabi.DTYPE_CODE[x.dtype], P, Q, x.stride(0), q.data_ptr(), q.stride(0),
                                                      scale.data_ptr(), absmax.data_ptr(), eps, int(stochastic_rounding), seed, offset,
                                                      _cabi.stream_of(x))
    _cabi.check(rc, "quantize_int8_colwise_given")
    return q, scale
