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

# `rng` (stochastic rounding only): a 0-dim int64 HOST tensor holding this call's Philox key, drawn by the Python wrappers below from
# torch's CPU generator (`torch.randint`, a node no compiler pass merges or reorders away).  It makes two SR calls on the same input
# differ in an ARGUMENT, so a tracing compiler's common-subexpression pass cannot fold them into one draw, it follows
# `torch.manual_seed`, and reading it costs no device synchronisation.
lib.define("quantize_int8(Tensor x, int dim, bool stochastic_rounding, float eps, Tensor? rng=None) -> (Tensor, Tensor)")
# Fused dual-direction pair: every X, W and dY of a mixed-precision Linear is quantized along BOTH dims
# per step (reference mixed_precision.py:100-108 with :154,:178,:186).  `quantize_int8_rowwise_colmax` is quantize_int8(x,
# dim=1) that also returns the fp32 absmax of every column from the same read; `quantize_int8_colwise_given` is
# quantize_int8(x, dim=0) for a caller who already holds that absmax (one read instead of two).
lib.define("quantize_int8_rowwise_colmax(Tensor x, bool stochastic_rounding, float eps, Tensor? rng=None) -> (Tensor, Tensor, Tensor)")
lib.define("quantize_int8_colwise_given(Tensor x, Tensor absmax, bool stochastic_rounding, float eps, Tensor? rng=None) -> (Tensor, Tensor)")


def draw_rng(stochastic_rounding: bool):
    """The `rng` argument of one op call: None without stochastic rounding, else a fresh 62-bit Philox key from torch's CPU generator."""
    return torch.randint(0, 1 << 62, (), dtype=torch.int64) if stochastic_rounding else None


def on_device(t: Tensor) -> bool:
    """True for tensors the device ops take.  One function, so that the CPU test-suite can route host tensors through the op path
    (with the oracle registered as the ops' CPU backend) and trace it under torch.compile without a GPU."""
    return t.is_cuda


def supports(x: Tensor, dim: int) -> bool:
    """Layouts the device op takes: 2-D of any strides, or N-D contiguous reducing the last dim."""
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        return False
    if x.ndim == 2:
        return True
    return x.ndim >= 1 and dim % x.ndim == x.ndim - 1 and x.is_contiguous()


def quantize_int8(x: Tensor, stochastic_rounding: bool = False, *, dim: int = -1, eps: float = 1e-12):
    """(int8 data with x's strides, scale with keepdim shape in x.dtype) -- see subclasses/int8.py."""
    return lib_ops.quantize_int8(x, dim, stochastic_rounding, eps, draw_rng(stochastic_rounding))


def _scale_shape(x: Tensor, dim: int):
    shape = list(x.shape)
    shape[dim % x.ndim] = 1
    return shape


@torch.library.impl(lib, "quantize_int8", "Meta")
def _quantize_int8_meta(x: Tensor, dim: int, stochastic_rounding: bool, eps: float, rng: Tensor | None = None):
    q = torch.empty_strided(x.shape, x.stride(), dtype=torch.int8, device=x.device) if x.ndim == 2 and _dense(x) else torch.empty_like(
        x, dtype=torch.int8, memory_format=torch.contiguous_format if x.ndim != 2 else torch.preserve_format)
    return q, x.new_empty(_scale_shape(x, dim))


@_cabi.device_impl(lib, "quantize_int8")
def _quantize_int8_hip(x: Tensor, dim: int, stochastic_rounding: bool, eps: float, rng: Tensor | None = None):
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
    seed, offset = _sr_args(stochastic_rounding, rng)
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
def _sr_args(stochastic_rounding: bool, rng: Tensor | None):
    """(Philox key, counter offset) of one stochastic-rounding call.  The key is the call's `rng` draw (see the schema comment); an op
    called directly without one draws here.  The values reach the kernel as launch arguments: a captured HIP graph replays the SAME
    noise on every replay -- do not capture stochastic-rounding steps."""
    if not stochastic_rounding:
        return 0, 0
    if rng is None:
        rng = draw_rng(True)
    return int(rng.item()) & 0xFFFFFFFFFFFFFFFF, 0


def supports_rowwise_colmax(x: Tensor) -> bool:
    """2-D device tensor with contiguous rows of at most 6144 (16-bit) / 3072 (fp32) elements whose pitch is a multiple of 16 bytes.
    Shape / stride / dtype only (no data_ptr): callable on fake tensors while torch.compile traces; a base address that is not
    16-byte aligned is handled inside the op (it re-packs the input first)."""
    if not (on_device(x) and x.ndim == 2 and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.numel() > 0):
        return False
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        return False
    esz = x.element_size()
    vec = 16 // esz
    C = x.shape[1]
    # mirrors rowcol_supported() of csrc/gnk_quantize.hip (checked against the library in tests/test_boundary_cpu.py)
    return (x.stride(0) * esz) % 16 == 0 and C % vec == 0 and (C // vec + 63) // 64 <= 12


def supports_colwise_given(x: Tensor) -> bool:
    """Shape / stride / dtype only, as supports_rowwise_colmax."""
    if not (on_device(x) and x.ndim == 2 and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.numel() > 0):
        return False
    esz = x.element_size()
    return x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.shape[1] % (16 // esz) == 0 and (x.stride(0) * esz) % 16 == 0


def _aligned16(x: Tensor) -> Tensor:
    """x itself, or a freshly allocated (hence aligned) copy when its base address is not a multiple of 16 bytes."""
    return x if x.data_ptr() % 16 == 0 else x.clone(memory_format=torch.contiguous_format)


def quantize_int8_rowwise_colmax(x: Tensor, stochastic_rounding: bool = False, *, eps: float = 1e-12):
    """(int8 data, scale (R, 1) in x.dtype, column absmax (C,) fp32): quantize_int8(x, dim=1) plus amax|x| over dim 0."""
    return lib_ops.quantize_int8_rowwise_colmax(x, stochastic_rounding, eps, draw_rng(stochastic_rounding))


def quantize_int8_colwise_given(x: Tensor, absmax: Tensor, stochastic_rounding: bool = False, *, eps: float = 1e-12):
    """quantize_int8(x, dim=0) given amax|x| over dim 0 (fp32, (C,)): (int8 data, scale (1, C) in x.dtype)."""
    return lib_ops.quantize_int8_colwise_given(x, absmax, stochastic_rounding, eps, draw_rng(stochastic_rounding))


@torch.library.impl(lib, "quantize_int8_rowwise_colmax", "Meta")
def _rowwise_colmax_meta(x: Tensor, stochastic_rounding: bool, eps: float, rng: Tensor | None = None):
    return torch.empty_like(x, dtype=torch.int8), x.new_empty((x.shape[0], 1)), x.new_empty((x.shape[1],), dtype=torch.float32)


@torch.library.impl(lib, "quantize_int8_colwise_given", "Meta")
def _colwise_given_meta(x: Tensor, absmax: Tensor, stochastic_rounding: bool, eps: float, rng: Tensor | None = None):
    return torch.empty_like(x, dtype=torch.int8), x.new_empty((1, x.shape[1]))


@_cabi.device_impl(lib, "quantize_int8_rowwise_colmax")
def _rowwise_colmax_hip(x: Tensor, stochastic_rounding: bool, eps: float, rng: Tensor | None = None):
    if not supports_rowwise_colmax(x):
        raise RuntimeError(f"gn_kernels::quantize_int8_rowwise_colmax: unsupported input (shape {tuple(x.shape)}, strides {x.stride()}, dtype {x.dtype})")
    x = _aligned16(x)
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
    seed, offset = _sr_args(stochastic_rounding, rng)
    with _cabi.device_of(x):
        rc = L.gnk_quantize_int8_rowwise_colmax(x.data_ptr(), code, R, C, x.stride(0), q.data_ptr(), q.stride(0), scale.data_ptr(),
                                                colmax.data_ptr(), eps, int(stochastic_rounding), seed, offset, ws.data_ptr(), ws_bytes,
                                                _cabi.stream_of(x))
    _cabi.check(rc, "quantize_int8_rowwise_colmax")
    return q, scale, colmax


@_cabi.device_impl(lib, "quantize_int8_colwise_given")
def _colwise_given_hip(x: Tensor, absmax: Tensor, stochastic_rounding: bool, eps: float, rng: Tensor | None = None):
    if not supports_colwise_given(x):
        raise RuntimeError(f"gn_kernels::quantize_int8_colwise_given: unsupported input (shape {tuple(x.shape)}, strides {x.stride()}, dtype {x.dtype})")
    x = _aligned16(x)
    P, Q = x.shape
    if not (absmax.is_cuda and absmax.dtype is torch.float32 and absmax.numel() == Q and absmax.is_contiguous()):
        raise RuntimeError("gn_kernels::quantize_int8_colwise_given: absmax must be a contiguous fp32 device tensor with one value per column")
    L = _cabi.load()
    q = torch.empty_like(x, dtype=torch.int8)
    if q.stride(1) != 1:
        q = torch.empty((P, Q), dtype=torch.int8, device=x.device)
    scale = torch.empty((1, Q), dtype=x.dtype, device=x.device)
    seed, offset = _sr_args(stochastic_rounding, rng)
    with _cabi.device_of(x):
        rc = L.gnk_quantize_int8_colwise_given_absmax(x.data_ptr(), _cabi.DTYPE_CODE[x.dtype], P, Q, x.stride(0), q.data_ptr(), q.stride(0),
                                                      scale.data_ptr(), absmax.data_ptr(), eps, int(stochastic_rounding), seed, offset,
                                                      _cabi.stream_of(x))
    _cabi.check(rc, "quantize_int8_colwise_given")
    return q, scale
