"""Device ops of the BitNet b1.58 weight path (csrc/gnk_bitnet.hip), SURVEY.md section 8f rank 4.

The reference keeps these as eager ATen chains inside subclasses/bitnet.py (`get_bitnet_scale` :110-112,
`quantize_bitnet_weight` :115-118, `_pack_i2_in_i8` :185-195, `_unpack_i2_in_i8` :198-201); here each is ONE HBM-bound
kernel, exposed additively under the `gn_kernels` namespace (opaque functional nodes for torch.compile).  Device tensors
only: there is no host implementation behind these ops.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _cabi
from ._lib import lib, lib_ops

lib.define("bitnet_absmean(Tensor x) -> Tensor")
lib.define("bitnet_quantize_weight(Tensor w, Tensor scale, float eps, bool packed) -> Tensor")
lib.define("bitnet_pack_i2(Tensor x) -> Tensor")
lib.define("bitnet_unpack_i2(Tensor x) -> Tensor")

_FLOATS = (torch.bfloat16, torch.float16, torch.float32)


def bitnet_absmean(x: Tensor) -> Tensor:
    """mean(|x|) over the whole tensor as a 0-dim fp32 tensor."""
    return lib_ops.bitnet_absmean(x)


def bitnet_quantize_weight(w: Tensor, scale: Tensor, eps: float = 1e-5, *, packed: bool = False) -> Tensor:
    """clip(round(w / max(scale, eps)), -1, 1) as int8 of w's shape, or (packed) four values per byte along the last dim."""
    return lib_ops.bitnet_quantize_weight(w, scale, eps, packed)


def bitnet_pack_i2(x: Tensor) -> Tensor:
    """(M, N) int8 -> (M, N / 4) int8, four 2-bit fields per byte, first value in the top bits."""
    return lib_ops.bitnet_pack_i2(x)


def bitnet_unpack_i2(x: Tensor) -> Tensor:
    """(M, N) packed int8 -> (M, 4 N) int8 with sign-extended 2-bit fields."""
    return lib_ops.bitnet_unpack_i2(x)


def _packed_shape(shape):
    assert len(shape) >= 1 and shape[-1] % 4 == 0, f"last dim of {tuple(shape)} is not a multiple of 4"
    return (*shape[:-1], shape[-1] // 4)


@torch.library.impl(lib, "bitnet_absmean", "Meta")
def _absmean_meta(x: Tensor):
    return x.new_empty((), dtype=torch.float32)


@torch.library.impl(lib, "bitnet_quantize_weight", "Meta")
def _quantize_weight_meta(w: Tensor, scale: Tensor, eps: float, packed: bool):
    return w.new_empty(_packed_shape(w.shape) if packed else w.shape, dtype=torch.int8)


@torch.library.impl(lib, "bitnet_pack_i2", "Meta")
def _pack_meta(x: Tensor):
    return x.new_empty(_packed_shape(x.shape))


@torch.library.impl(lib, "bitnet_unpack_i2", "Meta")
def _unpack_meta(x: Tensor):
    return x.new_empty((*x.shape[:-1], x.shape[-1] * 4))


def _need(cond: bool, op: str, what: str):
    if not cond:
        raise RuntimeError(f"gn_kernels::{op}: {what}")


@torch.library.impl(lib, "bitnet_absmean", "CUDA")
def _absmean_hip(x: Tensor):
    _need(x.dtype in _FLOATS and x.is_contiguous() and x.numel() > 0, "bitnet_absmean",
          f"needs a non-empty contiguous bf16 / fp16 / fp32 tensor (shape {tuple(x.shape)}, strides {x.stride()}, dtype {x.dtype})")
    L = _cabi.load()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws_bytes = L.gnk_bitnet_absmean_workspace_bytes()
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with _cabi.device_of(x):
        rc = L.gnk_bitnet_absmean(x.data_ptr(), _cabi.DTYPE_CODE[x.dtype], x.numel(), out.data_ptr(), ws.data_ptr(), ws_bytes,
                                  _cabi.stream_of(x))
    _cabi.check(rc, "bitnet_absmean")
    return out


@torch.library.impl(lib, "bitnet_quantize_weight", "CUDA")
def _quantize_weight_hip(w: Tensor, scale: Tensor, eps: float, packed: bool):
    _need(w.dtype in _FLOATS and w.is_contiguous(), "bitnet_quantize_weight",
          f"needs a contiguous bf16 / fp16 / fp32 weight (shape {tuple(w.shape)}, strides {w.stride()}, dtype {w.dtype})")
    _need(scale.dtype is torch.float32 and scale.numel() == 1 and scale.device == w.device, "bitnet_quantize_weight",
          "scale must be one fp32 value on the weight's device")
    out = torch.empty(_packed_shape(w.shape) if packed else w.shape, dtype=torch.int8, device=w.device)
    if w.numel() == 0:
        return out
    with _cabi.device_of(w):
        rc = _cabi.load().gnk_bitnet_quantize_weight(w.data_ptr(), _cabi.DTYPE_CODE[w.dtype], w.numel(), scale.data_ptr(), eps,
                                                     out.data_ptr(), int(packed), _cabi.stream_of(w))
    _cabi.check(rc, "bitnet_quantize_weight")
    return out


@torch.library.impl(lib, "bitnet_pack_i2", "CUDA")
def _pack_hip(x: Tensor):
    _need(x.dtype is torch.int8 and x.is_contiguous(), "bitnet_pack_i2", "needs a contiguous int8 tensor")
    out = torch.empty(_packed_shape(x.shape), dtype=torch.int8, device=x.device)
    if x.numel():
        with _cabi.device_of(x):
            rc = _cabi.load().gnk_bitnet_pack_i2(x.data_ptr(), x.numel(), out.data_ptr(), _cabi.stream_of(x))
        _cabi.check(rc, "bitnet_pack_i2")
    return out


@torch.library.impl(lib, "bitnet_unpack_i2", "CUDA")
def _unpack_hip(x: Tensor):
    _need(x.dtype is torch.int8 and x.is_contiguous() and x.ndim >= 1, "bitnet_unpack_i2", "needs a contiguous int8 tensor")
    out = torch.empty((*x.shape[:-1], x.shape[-1] * 4), dtype=torch.int8, device=x.device)
    if x.numel():
        with _cabi.device_of(x):
            rc = _cabi.load().gnk_bitnet_unpack_i2(x.data_ptr(), x.numel(), out.data_ptr(), _cabi.stream_of(x))
        _cabi.check(rc, "bitnet_unpack_i2")
    return out
