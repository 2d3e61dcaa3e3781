"""BitNet b1.58 training weight and its 2-bit FSDP all-gather container (API of reference subclasses/bitnet.py).

`F.linear` on a `BitNetTrainingLinearWeight` quantizes the activation row-wise to INT8 (eps 1e-5), the weight to
{-1, 0, +1} with ONE tensor-wise abs-mean scale, and runs `scaled_mm` with a row scale and a scalar column scale --
the `COL_SCALE_SCALAR` form of the reference's kernel (triton_mm.py:405).  Backward stays in the activation dtype and
uses the saved INT8 copies (reference :146-168).

Against the reference:
  * the scalar scale reaches `scaled_mm` as a (1, 1) tensor; the reference passes the 0-dim tensor and trips the
    wrapper's `ndim == 2` check at this snapshot (bitnet.py:133-140 vs triton_mm.py:362);
  * device tensors go through hand-written kernels (gn_kernels/bitnet.py): abs-mean, ternary quantize, and -- for the
    FSDP2 pre-all-gather hook -- quantize fused with the 2-bit packing, so the bf16 shard is read once and 0.25 B per
    element is written (the reference writes an int8 copy and then packs it with seven more ATen kernels);
  * host tensors take the same arithmetic in ATen (a module may be converted before `.cuda()`); device tensors never do.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F
import torch.utils._pytree as pytree
from torch import Tensor, nn

from gn_kernels import bitnet as _bop
from gn_kernels import scaled_mm

from .int8 import quantize_int8
from .mixed_precision import _REWRAP_OPS

aten = torch.ops.aten
_ACT_EPS = 1e-5      # activation quantizer's divisor floor (reference :129, :274)


def get_bitnet_scale(x: Tensor) -> Tensor:
    """Tensor-wise abs-mean, always fp32 and 0-dim (reference :110-112)."""
    if x.is_cuda:
        return _bop.bitnet_absmean(x if x.is_contiguous() else x.contiguous())
    return x.float().abs().mean()


def quantize_bitnet_weight(w: Tensor, scale: Tensor, eps: float = 1e-5) -> Tensor:
    """{-1, 0, +1} as int8: clip(round(w / max(scale, eps)), -1, 1) (reference :115-118)."""
    if w.is_cuda:
        return _bop.bitnet_quantize_weight(w if w.is_contiguous() else w.contiguous(), scale.to(device=w.device, dtype=torch.float32), eps)
    return (w.float() / scale.clip(eps)).round().clip(-1, 1).to(torch.int8)


def _pack_i2_in_i8(x: Tensor) -> Tensor:
    """(M, N) int8 -> (M, N / 4): bytes [a, b, c, d] become aabbccdd (reference :185-195)."""
    if x.is_cuda:
        return _bop.bitnet_pack_i2(x if x.is_contiguous() else x.contiguous())
    return (x[:, ::4] << 6) | ((x[:, 1::4] & 3) << 4) | ((x[:, 2::4] & 3) << 2) | (x[:, 3::4] & 3)


def _unpack_i2_in_i8(x: Tensor) -> Tensor:
    """(M, N) packed -> (M, 4 N) int8; int8 shifts sign-extend each 2-bit field (reference :198-201)."""
    if x.is_cuda:
        return _bop.bitnet_unpack_i2(x if x.is_contiguous() else x.contiguous())
    return torch.stack([x >> 6, x << 2 >> 6, x << 4 >> 6, x << 6 >> 6], dim=-1).view(x.shape[0], -1)


def _quantize_and_pack(w: Tensor, scale: Tensor, eps: float = 1e-5) -> Tensor:
    """quantize_bitnet_weight + _pack_i2_in_i8 in one pass over the weight."""
    if w.is_cuda:
        return _bop.bitnet_quantize_weight(w if w.is_contiguous() else w.contiguous(), scale.to(device=w.device, dtype=torch.float32), eps, packed=True)
    return _pack_i2_in_i8(quantize_bitnet_weight(w, scale, eps))


def _all_reduce_mean(t: Tensor) -> None:
    """Average over the default group.  RCCL has a native AVG; gloo (CPU tests) does not."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(dist.get_world_size())


class BitNetTrainingLinearWeight(Tensor):
    """High-precision master weight whose F.linear runs as INT8 activations x ternary weights."""

    @staticmethod
    @torch._dynamo.disable
    def __new__(cls, data: Tensor):
        return Tensor._make_wrapper_subclass(cls, data.shape, dtype=data.dtype, device=data.device)

    @torch._dynamo.disable
    def __init__(self, data: Tensor):
        self._data = data

    def __tensor_flatten__(self):
        return ["_data"], []

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size=None, outer_stride=None):
        return cls(tensor_data_dict["_data"], *tensor_attributes)

    def __repr__(self):
        return f"{type(self).__name__}(data={self._data})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is F.linear:
            return _BitNetTrainingLinear.apply(*args, **(kwargs or {}))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    @classmethod
    def __torch_dispatch__(cls, func, types, args, kwargs):
        inner = lambda w: w._data  # noqa: E731
        out = func(*pytree.tree_map_only(cls, inner, args), **pytree.tree_map_only(cls, inner, kwargs or {}))
        if func is aten.copy_.default:
            return args[0]                       # in-place: hand back the original wrapper
        if func in _REWRAP_OPS:                  # views / copies of the parameter stay wrapped (reference :62-76)
            return pytree.tree_map_only(Tensor, cls, out)
        return out

    # FSDP2 extension points (reference :83-107): all-gather the weight as 2-bit codes + one global scale.  The reference's
    # (mesh, module=None, mp_policy=None) signature is the PyTorch 2.5 one; this image's FSDP2 (2.10) accepts exactly 1 or 5
    # parameters, so the current five are taken, all but `mesh` optional.
    def fsdp_pre_all_gather(self, mesh, outer_size=None, outer_stride=None, module=None, mp_policy=None):
        scale = get_bitnet_scale(self._data)     # abs-mean of the local shard, fp32
        _all_reduce_mean(scale)                  # equal shards: mean of shard means = tensor-wise mean
        return (_quantize_and_pack(self._data, scale),), (scale,)

    def fsdp_post_all_gather(self, all_gather_outputs: Tuple[Tensor, ...], metadata: Any, param_dtype: torch.dtype, *,
                             out: Optional[Tensor] = None):
        (data_i2,) = all_gather_outputs
        (scale,) = metadata
        scale = scale.to(param_dtype)
        if out is not None:
            assert isinstance(out, BitNetPacked2bitLinearWeight)
            out.scale = scale
            return
        return BitNetPacked2bitLinearWeight(data_i2, scale), all_gather_outputs


def _bitnet_forward(ctx, input: Tensor, weight_i8: Tensor, tensor_scale: Tensor, out_features: int, bias: Tensor | None):
    """Shared by both weight classes: INT8 activations x ternary weights on the INT8 MFMA path."""
    lead = input.shape[:-1]
    x2d = input.reshape(-1, weight_i8.shape[1])
    input_i8, row_scale = quantize_int8(x2d, eps=_ACT_EPS)
    out = scaled_mm(input_i8, weight_i8.T, row_scale, tensor_scale.reshape(1, 1)).view(*lead, out_features)
    return (out + bias if bias is not None else out), input_i8, row_scale


def _bitnet_backward(ctx, grad_output: Tensor, input_i8: Tensor, row_scale: Tensor, weight_i8: Tensor, tensor_scale: Tensor):
    """Reference :146-168 / :289-311: activation-dtype matmuls against the saved quantized copies."""
    grad_input = grad_weight = grad_bias = None
    lead = grad_output.shape[:-1]
    go2d = grad_output.reshape(-1, weight_i8.shape[0])
    if ctx.needs_input_grad[0]:
        grad_input = ((go2d @ weight_i8.to(go2d.dtype)) * tensor_scale).view(*lead, weight_i8.shape[1])
    if ctx.needs_input_grad[1]:
        grad_weight = go2d.T @ (input_i8 * row_scale.view(-1, 1))      # the DEQUANTIZED activation, as the reference
    if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2]:
        grad_bias = go2d.sum(0)
    return (grad_input, grad_weight, grad_bias)[:len(ctx.needs_input_grad)]     # F.linear(x, w) passes no bias slot at all


class _BitNetTrainingLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input: Tensor, weight: BitNetTrainingLinearWeight, bias: Tensor | None = None):
        # fp32 scale for the weight quantization, the weight's dtype for the matmul and backward (reference :131-135)
        scale_f32 = get_bitnet_scale(weight._data)
        weight_i8 = quantize_bitnet_weight(weight._data, scale_f32)
        tensor_scale = scale_f32.to(weight.dtype)
        out, input_i8, row_scale = _bitnet_forward(ctx, input, weight_i8, tensor_scale, weight.shape[0], bias)
        ctx.save_for_backward(input_i8, row_scale, weight_i8, tensor_scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        return _bitnet_backward(ctx, grad_output, *ctx.saved_tensors)


def convert_bitnet(module: nn.Module):
    """Wrap every nn.Linear weight under `module` (in place).  RMSNorm placement is the caller's business (reference :171-182)."""
    if isinstance(module, nn.Linear):
        module.weight = nn.Parameter(BitNetTrainingLinearWeight(module.weight.detach()), requires_grad=module.weight.requires_grad)
    else:
        for child in module.children():
            convert_bitnet(child)
    return module


class BitNetPacked2bitLinearWeight(Tensor):
    """(M, N / 4) bytes of 2-bit codes + one scale: what FSDP2 all-gathers in place of the bf16 weight (reference :206-263)."""

    @staticmethod
    @torch._dynamo.disable
    def __new__(cls, int_data: Tensor, scale: Tensor):
        rows, packed_cols = int_data.shape
        return Tensor._make_wrapper_subclass(cls, (rows, packed_cols * 4), dtype=scale.dtype, device=scale.device)

    @torch._dynamo.disable
    def __init__(self, int_data: Tensor, scale: Tensor):
        assert int_data.dtype is torch.int8
        assert scale.shape == ()
        self.int_data = int_data
        self.scale = scale

    def __tensor_flatten__(self):
        return ["int_data", "scale"], []

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size=None, outer_stride=None):
        return cls(tensor_data_dict["int_data"], tensor_data_dict["scale"], *tensor_attributes)

    def __repr__(self):
        return f"{type(self).__name__}(data={self.dequantize()})"

    def dequantize(self, out_dtype=None):
        out = _unpack_i2_in_i8(self.int_data) * self.scale
        return out if out_dtype is None else out.to(out_dtype)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is F.linear:
            return _BitNetPacked2bitLinear.apply(*args, **(kwargs or {}))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    @classmethod
    def __torch_dispatch__(cls, func, types, args, kwargs):
        kwargs = kwargs or {}
        if func in (aten.detach.default, aten.clone.default):
            w = args[0]
            return cls(func(w.int_data, *args[1:], **kwargs), func(w.scale, *args[1:], **kwargs))
        if func is aten.as_strided.default:      # FSDP2 re-views the gathered parameter: same bytes, same scale
            return cls(args[0].int_data, args[0].scale)
        raise NotImplementedError(f"{cls.__name__} dispatch: attempting to run {func}, this is not supported")


class _BitNetPacked2bitLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input: Tensor, weight: BitNetPacked2bitLinearWeight, bias: Tensor | None = None):
        weight_i2, tensor_scale = weight.int_data, weight.scale
        out, input_i8, row_scale = _bitnet_forward(ctx, input, _unpack_i2_in_i8(weight_i2), tensor_scale, weight.shape[0], bias)
        ctx.save_for_backward(input_i8, row_scale, weight_i2, tensor_scale)      # the PACKED weight is what is kept
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input_i8, row_scale, weight_i2, tensor_scale = ctx.saved_tensors
        return _bitnet_backward(ctx, grad_output, input_i8, row_scale, _unpack_i2_in_i8(weight_i2), tensor_scale)
