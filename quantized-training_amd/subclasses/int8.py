"""INT8 quantizer and INT8-stored Linear weight (API of reference subclasses/int8.py).

`quantize_int8` keeps the reference's signature and numerics (int8.py:12-26).  For device tensors it
is ONE hand-written HBM-bound kernel (`gn_kernels::quantize_int8`); for host tensors it is the same
arithmetic in ATen ops, because model conversion (`from_float`) may run before `.cuda()`.  Device
tensors never take the ATen route: a missing libgnk_hip.so raises.

`Int8LinearWeight` is a wrapper tensor subclass holding (int8 data, per-row scale).  Behavioural
notes against the reference:
  * the dynamic-activation forward passes the weight scale as (1, N); the reference passes (N, 1)
    and trips `scaled_mm`'s shape check at this snapshot;
  * the optimizer-step re-quantization (`copy_` of a high-precision tensor) uses stochastic
    rounding exactly as int8.py:119-122.
"""
from __future__ import annotations

from typing import Literal, NamedTuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from gn_kernels import quantize as _qop
from gn_kernels import scaled_mm

aten = torch.ops.aten


def _quantize_int8_aten(x: Tensor, stochastic_rounding: bool, dim: int, eps: float):
    xf = x.to(torch.float32)
    scale = xf.abs().amax(dim, keepdim=True) / 127
    xf = xf / scale.clamp_min(eps)
    xf = torch.floor(xf + torch.rand_like(xf)) if stochastic_rounding else torch.round(xf)
    return xf.clamp(-128, 127).to(torch.int8), scale.to(x.dtype)


@torch.no_grad()
def quantize_int8(tensor: Tensor, stochastic_rounding: bool = False, *, dim: int = -1, eps: float = 1e-12):
    """Absmax symmetric INT8 quantization along `dim`: returns (int8 data, scale in tensor.dtype).

    scale = amax|x| / 127 (fp32), data = clip(round(x / max(scale, eps)), -128, 127); with
    `stochastic_rounding` the rounding is floor(x / scale + U[0, 1)).
    """
    if _qop.on_device(tensor) and _qop.supports(tensor, dim):
        return _qop.quantize_int8(tensor, stochastic_rounding, dim=dim, eps=eps)
    if _qop.on_device(tensor) and tensor.ndim >= 1:
        # N-D device tensor reduced along an inner dim: move `dim` last, quantize, move back
        moved = tensor.movedim(dim, -1).contiguous()
        q, s = _qop.quantize_int8(moved, stochastic_rounding, dim=-1, eps=eps)
        return q.movedim(-1, dim), s.movedim(-1, dim)
    return _quantize_int8_aten(tensor, stochastic_rounding, dim, eps)


class Int8QTConfig(NamedTuple):
    activation: Literal["none", "int8", "int8_sr"] = "none"


class Int8LinearWeight(Tensor):
    """nn.Linear weight stored as int8 rows + one scale per output feature."""

    @staticmethod
    @torch._dynamo.disable
    def __new__(cls, int_data: Tensor, scale: Tensor, config: Int8QTConfig):
        return Tensor._make_wrapper_subclass(cls, int_data.shape, dtype=scale.dtype, device=int_data.device)

    @torch._dynamo.disable
    def __init__(self, int_data: Tensor, scale: Tensor, config: Int8QTConfig):
        assert int_data.dtype is torch.int8
        assert int_data.ndim == 2
        assert scale.ndim == 2
        self.int_data = int_data
        self.scale = scale
        self.config = config

    def __tensor_flatten__(self):
        return ["int_data", "scale"], [self.config]

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size=None, outer_stride=None):
        return cls(tensor_data_dict["int_data"], tensor_data_dict["scale"], *tensor_attributes)

    @classmethod
    def from_float(cls, tensor: Tensor, *, config: Int8QTConfig = Int8QTConfig()):
        int_data, scale = quantize_int8(tensor.detach())
        out = cls(int_data, scale, config)
        out.requires_grad_(tensor.requires_grad)
        return out

    def dequantize(self) -> Tensor:
        return self.int_data * self.scale.view(-1, 1)

    def __repr__(self):
        return (f"{type(self).__name__}(shape={tuple(self.shape)}, config={self.config}, dtype={self.dtype}, "
                f"device={self.device}, requires_grad={self.requires_grad})")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is F.linear:
            return _Int8Linear.apply(*args, **(kwargs or {}))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    @classmethod
    def __torch_dispatch__(cls, func, types, args, kwargs):
        handler = _INT8_DISPATCH.get(func)
        if handler is None:
            raise NotImplementedError(f"{cls.__name__} dispatch: attempting to run {func}, this is not supported")
        return handler(cls, func, args, kwargs or {})


# ---- aten-level behaviour of Int8LinearWeight (reference int8.py:86-135), one handler per family
def _i8_rewrap(cls, func, args, kwargs):
    w = args[0]
    return cls(func(w.int_data, *args[1:], **kwargs), func(w.scale, *args[1:], **kwargs), w.config)


def _i8_to_copy(cls, func, args, kwargs):
    w = args[0]
    device, dtype = kwargs.get("device"), kwargs.get("dtype")
    return cls(w.int_data.to(device=device), w.scale.to(device=device, dtype=dtype), w.config)


def _i8_zeros_like(cls, func, args, kwargs):
    # optimizer state must be an ordinary tensor, not an INT8 weight
    w = args[0]
    return torch.zeros(w.shape, dtype=kwargs.get("dtype", w.dtype), device=kwargs.get("device", w.device))


def _i8_dequant_binary(cls, func, args, kwargs):
    return func(*[a.dequantize() if isinstance(a, cls) else a for a in args], **kwargs)


def _i8_copy_(cls, func, args, kwargs):
    dst, src = args[0], args[1]
    if isinstance(dst, cls) and isinstance(src, cls):
        dst.int_data.copy_(src.int_data)
        dst.scale.copy_(src.scale)
    elif isinstance(dst, cls):
        # high-precision update written back into INT8 storage: stochastic rounding keeps it unbiased
        int_data, scale = quantize_int8(src, stochastic_rounding=True)
        dst.int_data.copy_(int_data)
        dst.scale.copy_(scale)
    else:
        dst.copy_(src.dequantize())
    return dst


def _i8_inplace_update(cls, func, args, kwargs):
    # optimizer step on the weight itself: update a dequantized copy, re-quantize into place
    w = args[0]
    return w.copy_(func(w.dequantize(), *args[1:], **kwargs))


_INT8_DISPATCH = {
    aten.detach.default: _i8_rewrap,
    aten.clone.default: _i8_rewrap,
    aten._to_copy.default: _i8_to_copy,
    aten.zeros_like.default: _i8_zeros_like,
    aten.sub.Tensor: _i8_dequant_binary,
    aten.mul.Tensor: _i8_dequant_binary,
    aten.copy_.default: _i8_copy_,
    aten.addcdiv_.default: _i8_inplace_update,
    aten.add_.Tensor: _i8_inplace_update,
    aten.mul_.Tensor: _i8_inplace_update,    # AdamW's decoupled weight decay (not handled by the reference)
}


class _Int8Linear(torch.autograd.Function):
    """F.linear for an Int8LinearWeight (reference int8.py:138-180): INT8 forward, bf16 backward."""

    @staticmethod
    def forward(ctx, input: Tensor, weight: Int8LinearWeight, bias: Tensor | None = None):
        ctx.save_for_backward(input, weight)
        ctx.bias = bias is not None

        if weight.config.activation == "none":
            # weight-only: cast the int8 rows on the fly, fold the scale into the output columns
            out = (input @ weight.int_data.T.to(input.dtype)) * weight.scale.T
        else:
            lead = input.shape[:-1]
            x2d = input.reshape(-1, weight.shape[1])
            x_i8, x_scale = quantize_int8(x2d, weight.config.activation == "int8_sr")
            out = scaled_mm(x_i8, weight.int_data.T, x_scale, weight.scale.T.contiguous())
            out = out.view(*lead, -1)

        return out + bias if bias is not None else out

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        # the weight scale runs along the reduction dim of both backward products, so they stay bf16
        grad_input = (grad_output * weight.scale.T) @ weight.int_data.to(grad_output.dtype)
        go2d = grad_output.reshape(-1, weight.shape[0])
        grad_weight = go2d.T @ input.reshape(-1, weight.shape[1])
        grad_bias = go2d.sum(0) if ctx.bias else None
        return grad_input, grad_weight, grad_bias


def convert_int8_quantized_training(module: nn.Module, *, config: Int8QTConfig = Int8QTConfig()):
    """Swap every nn.Linear weight under `module` for an Int8LinearWeight (in place)."""
    if isinstance(module, nn.Linear):
        module.weight = nn.Parameter(Int8LinearWeight.from_float(module.weight, config=config),
                                     requires_grad=module.weight.requires_grad)
    else:
        for child in module.children():
            convert_int8_quantized_training(child, config=config)
    return module
