"""INT8 mixed-precision training weight (API of reference subclasses/mixed_precision.py).

The master weight stays bf16; `F.linear` on a `MixedPrecisionLinearWeight` is rerouted to an
autograd function whose three matmuls (output, grad-input, grad-weight) each optionally run as
dynamic row-wise INT8 quantization + scaled INT8 MFMA GEMM (reference :146-193).

MI355X-first difference: the reference quantizes `A` and `B.T` along dim 1 and then calls
`.contiguous()` on both (:100-108), i.e. it materialises transposes so that its Triton kernel always
sees row-major x column-major.  Here the operands are quantized IN PLACE OF LAYOUT -- the
strided-reduction quantize kernel keeps the tensor's memory order -- and the GEMM consumes the
M/N-contiguous operands through its transposed LDS read path, so forward, grad-input and
grad-weight run as A@B.T, A@B and A.T@B without any transpose copy.  The arithmetic (per-row /
per-column absmax scales, exact int32 accumulation, (acc * row) * col epilogue) is unchanged, so
results are bit-identical to the reference's.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn.functional as F
import torch.utils._pytree as pytree
from torch import Tensor, nn

from gn_kernels import scaled_mm

from .int8 import quantize_int8

aten = torch.ops.aten


class MixedPrecisionConfig(NamedTuple):
    output: bool = True
    grad_input: bool = True
    grad_weight: bool = True
    dtype: str = "int8"
    stochastic_rounding: bool = False


# aten ops whose result is re-wrapped in the subclass (views / copies of the parameter itself);
# every other op sees, and returns, plain tensors (reference :80-97)
_REWRAP_OPS = frozenset({
    aten.t.default,
    aten.detach.default,
    aten.empty_like.default,
    aten.new_zeros.default,
    aten.slice.Tensor,
    aten.view.default,
    aten.as_strided.default,
    aten._to_copy.default,
    aten._pin_memory.default,
    aten.split.Tensor,
    aten.clone.default,
})


class MixedPrecisionLinearWeight(Tensor):
    """bf16 master weight whose F.linear runs in INT8."""

    @staticmethod
    @torch._dynamo.disable
    def __new__(cls, data: Tensor, config: MixedPrecisionConfig):
        return Tensor._make_wrapper_subclass(cls, data.shape, dtype=data.dtype, device=data.device)

    @torch._dynamo.disable
    def __init__(self, data: Tensor, config: MixedPrecisionConfig):
        self._data = data
        self.config = config

    def __tensor_flatten__(self):
        return ["_data"], [self.config]

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size=None, outer_stride=None):
        return cls(tensor_data_dict["_data"], *tensor_attributes)

    def __repr__(self):
        return f"{type(self).__name__}(data={self._data}, config={self.config})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is F.linear:
            return _Int8MixedPrecisionLinear.apply(*args, **(kwargs or {}))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    @classmethod
    def __torch_dispatch__(cls, func, types, args, kwargs):
        seen: list[MixedPrecisionConfig] = []

        def unwrap(w):
            if seen:
                assert w.config == seen[0]
            else:
                seen.append(w.config)
            return w._data

        out = func(*pytree.tree_map_only(cls, unwrap, args), **pytree.tree_map_only(cls, unwrap, kwargs or {}))
        if func is aten.copy_.default:
            return args[0]                       # in-place: hand back the original wrapper
        if func in _REWRAP_OPS:
            return pytree.tree_map_only(Tensor, lambda t: cls(t, seen[0]), out)
        return out


def _dynamic_int8_mm(A: Tensor, B: Tensor, sr: bool) -> Tensor:
    """(M, K) @ (K, N) with A quantized per row and B per column, both along K (reference :100-108).
    No `.contiguous()`: each int8 operand keeps its source layout and `scaled_mm` takes any strides."""
    A_i8, row_scale = quantize_int8(A, sr, dim=1)    # scale (M, 1)
    B_i8, col_scale = quantize_int8(B, sr, dim=0)    # scale (1, N)
    return scaled_mm(A_i8, B_i8, row_scale, col_scale)


def _dynamic_mm(A: Tensor, B: Tensor, sr: bool, dtype: str) -> Tensor:
    if dtype == "int8":
        return _dynamic_int8_mm(A, B, sr)
    if dtype == "int4":
        raise NotImplementedError("MixedPrecisionConfig(dtype='int4') needs an INT4 MFMA, which CDNA4 does not have")
    raise ValueError(dtype)


class _Int8MixedPrecisionLinear(torch.autograd.Function):
    @staticmethod
    def forward(input: Tensor, weight: MixedPrecisionLinearWeight, bias: Tensor | None = None):
        cfg = weight.config
        if cfg.output:
            lead = input.shape[:-1]
            x2d = input.reshape(-1, weight.shape[1])
            out = _dynamic_mm(x2d, weight._data.T, cfg.stochastic_rounding, cfg.dtype)      # A @ B.T
            out = out.view(*lead, weight.shape[0])
        else:
            out = input @ weight.T
        return out + bias if bias is not None else out

    @staticmethod
    def setup_context(ctx, inputs, output):
        input, weight, bias = inputs
        ctx.config = weight.config
        ctx.save_for_backward(input, weight._data)    # bf16 tensors; int8 copies are re-derived in backward
        ctx.bias = bias is not None

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        cfg = ctx.config
        grad_input = grad_weight = grad_bias = None

        lead = grad_output.shape[:-1]
        go2d = grad_output.reshape(-1, weight.shape[0])
        x2d = input.reshape(-1, weight.shape[1])

        if ctx.needs_input_grad[0]:
            if cfg.grad_input:
                grad_input = _dynamic_mm(go2d, weight, cfg.stochastic_rounding, cfg.dtype)           # A @ B
            else:
                grad_input = go2d @ weight
            grad_input = grad_input.view(*lead, weight.shape[1])

        if ctx.needs_input_grad[1]:
            if cfg.grad_weight:
                grad_weight = _dynamic_mm(x2d.T, go2d, cfg.stochastic_rounding, cfg.dtype).T          # (A.T @ B).T
            else:
                grad_weight = go2d.T @ x2d

        if ctx.needs_input_grad[2] and ctx.bias:
            grad_bias = go2d.sum(0)

        return grad_input, grad_weight, grad_bias


def convert_mixed_precision(module: nn.Module, *, config: MixedPrecisionConfig = MixedPrecisionConfig()):
    """Wrap every nn.Linear weight under `module` in a MixedPrecisionLinearWeight (in place)."""
    if isinstance(module, nn.Linear):
        module.weight = nn.Parameter(MixedPrecisionLinearWeight(module.weight.detach(), config=config),
                                     requires_grad=module.weight.requires_grad)
    else:
        for child in module.children():
            convert_mixed_precision(child, config=config)
    return module
