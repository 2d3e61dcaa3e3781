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

import threading
import weakref
from typing import NamedTuple

import torch
import torch.nn.functional as F
import torch.utils._pytree as pytree
from torch import Tensor, nn

from gn_kernels import quantize as _qop
from gn_kernels import scaled_mm, scaled_mm_t

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

    # FSDP2 (llm_pretrain.py:134-139 `fully_shard(layer)`).  The reference defines no all-gather hooks for this class; FSDP2's default path
    # then rebuilds the unsharded parameter as a PLAIN tensor from its all-gather buffer (torch's _fsdp_param.py:init_unsharded_param), and
    # every F.linear silently runs in bf16.  With the two hooks the all-gathered bf16 weight comes back wrapped, so the sharded model runs
    # the same INT8 matmuls as the unsharded one (tests/test_subclasses_cpu.py, gloo, world size 2: bit-equal outputs).
    def fsdp_pre_all_gather(self, mesh, outer_size=None, outer_stride=None, module=None, mp_policy=None):
        data = self._data
        if mp_policy is not None and getattr(mp_policy, "param_dtype", None) is not None:
            data = data.to(mp_policy.param_dtype)
        return (data,), (self.config,)

    def fsdp_post_all_gather(self, all_gather_outputs, metadata, param_dtype, *, out=None):
        (data,) = all_gather_outputs
        (config,) = metadata
        if out is not None:                      # later all-gathers write into the storage `out._data` already views
            assert isinstance(out, MixedPrecisionLinearWeight)
            return
        return MixedPrecisionLinearWeight(data, config), all_gather_outputs

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


# ---------------------------------------------------------------------------------------------------------------------
# Quantization plan of one Linear (device tensors).  The reference quantizes X, W and dY along BOTH dims every step,
# each time from scratch (:100-108 called from :154, :178, :186), and q_proj / k_proj / v_proj (gate / up) re-quantize
# the very same activation; under torch.compile (llm_pretrain.py:184) the compiler's common-subexpression pass merges
# those duplicates.  Identical arithmetic on identical inputs either way, so results do not change:
#   * for large tensors the row-wise pass also collects the fp32 column absmax (gn_kernels::quantize_int8_rowwise_colmax),
#     which backward hands to the column-wise pass (gn_kernels::quantize_int8_colwise_given): one read instead of two.
#     Every decision is taken from shapes, strides and dtypes only, so the plan traces under torch.compile;
#   * EAGER MODE ONLY (never while a compiler traces: the traced graph is de-duplicated by the compiler itself): a small
#     cache returns the int8 copy of an ACTIVATION that was already quantized along that dim.  An entry is keyed by the
#     identity of the tensor OBJECT handed to F.linear (weak reference + version counter), so it dies with the
#     activation, pins nothing but its int8 copy, and is only made for tensors produced inside an autograd graph
#     (`grad_fn` set): those are fresh objects every forward, so no entry outlives a step, and a leaf the user may rewrite
#     through `.data` between two forwards is never cached.  Off under stochastic rounding (every call draws its own
#     random numbers), for inference tensors (no version counter) and for weights (the optimizer updates a wrapped weight
#     through __torch_dispatch__, below the layer that bumps version counters).
_FUSED_MIN_NUMEL = 48 * 1024 * 1024   # measured (tools/fused_quant_probe.py): the fused pair wins from ~32768 x 2048 up
_CACHE_SLOTS = 8                      # int8 copies alive at once (LRU); q/k/v and gate/up need 1-2
_cache_enabled = True
_fused_enabled = True
_cache: dict = {}                     # id(source) -> [weakref(source), version, {kind: value}]
_cache_lock = threading.RLock()       # forward runs on the caller's thread, backward on the autograd engine's; re-entrant: a GC pass inside
                                      # _cache_put can finalize a cached source tensor, whose weakref callback (_gone) takes the lock on the same thread


def set_quantize_plan(*, cache: bool | None = None, fused: bool | None = None) -> tuple[bool, bool]:
    """A/B switches (both on by default): the duplicate-quantization cache and the fused dual-direction kernels."""
    global _cache_enabled, _fused_enabled
    old = (_cache_enabled, _fused_enabled)
    if cache is not None:
        _cache_enabled = cache
        clear_quantize_cache()
    if fused is not None:
        _fused_enabled = fused
    return old


def clear_quantize_cache() -> None:
    """Drop the cached int8 copies."""
    with _cache_lock:
        _cache.clear()


def _cacheable(src: Tensor | None, sr: bool) -> bool:
    if src is None or sr or not _cache_enabled or torch.compiler.is_compiling():
        return False
    return src.grad_fn is not None and not src.is_inference()


def _cache_get(kind: str, src: Tensor | None, sr: bool):
    if not _cacheable(src, sr):
        return None
    with _cache_lock:
        ent = _cache.get(id(src))
        if ent is None:
            return None
        if ent[0]() is not src or ent[1] != src._version:     # the id was recycled, or the tensor was written since
            del _cache[id(src)]
            return None
        return ent[2].get(kind)


def _cache_put(kind: str, src: Tensor | None, sr: bool, value) -> None:
    if not _cacheable(src, sr):
        return
    key = id(src)

    def _gone(_ref, key=key):
        with _cache_lock:
            ent = _cache.get(key)
            if ent is not None and ent[0] is _ref:
                del _cache[key]

    with _cache_lock:
        ent = _cache.get(key)
        if ent is None or ent[0]() is not src or ent[1] != src._version:
            while len(_cache) >= _CACHE_SLOTS:
                _cache.pop(next(iter(_cache)))               # dicts keep insertion order: drop the oldest
            ent = _cache[key] = [weakref.ref(src, _gone), src._version, {}]
        ent[2][kind] = value


def _plan_ok(*ts: Tensor) -> bool:
    return all(_qop.on_device(t) and t.ndim == 2 and t.stride(1) == 1 and t.numel() > 0 for t in ts)


def _q_rows(x: Tensor, sr: bool, want_colmax: bool, src: Tensor | None = None):
    """quantize_int8(x, dim=1) -> (int8, scale (R, 1), fp32 column absmax or None).  `src`: the tensor object x was reshaped from
    (cache identity); None = do not cache (weights)."""
    hit = _cache_get("row", src, sr)
    if hit is not None and (hit[2] is not None or not want_colmax):
        return hit
    if want_colmax and _fused_enabled and x.numel() >= _FUSED_MIN_NUMEL and _qop.supports_rowwise_colmax(x):
        out = _qop.quantize_int8_rowwise_colmax(x, sr)
    else:
        out = (*quantize_int8(x, sr, dim=1), None)
    _cache_put("row", src, sr, out)
    return out


def _q_cols(x: Tensor, sr: bool, colmax: Tensor | None, src: Tensor | None = None):
    """quantize_int8(x, dim=0) -> (int8 in x's layout, scale (1, C)); `colmax` (if known) saves the absmax pass."""
    hit = _cache_get("col", src, sr)
    if hit is not None:
        return hit
    if colmax is not None and _qop.supports_colwise_given(x):
        out = _qop.quantize_int8_colwise_given(x, colmax, sr)
    else:
        out = quantize_int8(x, sr, dim=0)
    _cache_put("col", src, sr, out)
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
    """reference :146-193.  Combined forward(ctx, ...) form (the reference uses forward + setup_context): the column
    absmax collected by forward's row-wise passes is saved for backward beside the bf16 tensors."""

    @staticmethod
    def forward(ctx, input: Tensor, weight: MixedPrecisionLinearWeight, bias: Tensor | None = None):
        cfg = weight.config
        w = weight._data
        x_cm = w_cm = None
        if cfg.output:
            lead = input.shape[:-1]
            x2d = input.reshape(-1, w.shape[1])
            if cfg.dtype == "int8" and _plan_ok(x2d, w):
                sr = cfg.stochastic_rounding
                xq, xs, x_cm = _q_rows(x2d, sr, cfg.grad_weight and ctx.needs_input_grad[1], input)
                wq, ws, w_cm = _q_rows(w, sr, cfg.grad_input and ctx.needs_input_grad[0])
                out = scaled_mm(xq, wq.T, xs, ws.T)                                                # A @ B.T
            else:
                out = _dynamic_mm(x2d, w.T, cfg.stochastic_rounding, cfg.dtype)
            out = out.view(*lead, w.shape[0])
        else:
            out = input @ w.T
        ctx.config = cfg
        # bf16 tensors; int8 copies are re-derived in backward (reference :160-165).  The two small fp32 column-absmax vectors (or None)
        # travel the same way, so that a tracing compiler sees them as saved tensors
        ctx.save_for_backward(input, w, x_cm, w_cm)
        ctx.bias = bias is not None
        return out + bias if bias is not None else out

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, x_cm, w_cm = ctx.saved_tensors
        cfg = ctx.config
        sr = cfg.stochastic_rounding
        grad_input = grad_weight = grad_bias = None

        lead = grad_output.shape[:-1]
        go2d = grad_output.reshape(-1, weight.shape[0])
        x2d = input.reshape(-1, weight.shape[1])
        want_gi = ctx.needs_input_grad[0]
        want_gw = ctx.needs_input_grad[1]
        plan = cfg.dtype == "int8" and _plan_ok(go2d, x2d, weight)
        g_cm = None

        if want_gi:
            if cfg.grad_input and plan:
                gq, gs, g_cm = _q_rows(go2d, sr, want_gw and cfg.grad_weight)
                wq, ws = _q_cols(weight, sr, w_cm)
                grad_input = scaled_mm(gq, wq, gs, ws)                                             # A @ B
            elif cfg.grad_input:
                grad_input = _dynamic_mm(go2d, weight, sr, cfg.dtype)
            else:
                grad_input = go2d @ weight
            grad_input = grad_input.view(*lead, weight.shape[1])

        if want_gw:
            if cfg.grad_weight and plan:
                xq, xs = _q_cols(x2d, sr, x_cm, input)           # == quantize_int8(x2d.T, dim=1), transposed
                gq0, gs0 = _q_cols(go2d, sr, g_cm)
                # reference :184-186 `_dynamic_mm(input.T, grad_output).T`: the same numbers, written straight into the weight's (N, K)
                # layout instead of a transposed view that AccumulateGrad would copy (gn_kernels/mm.py:scaled_mm_t)
                grad_weight = scaled_mm_t(xq.T, gq0, xs.T, gs0)                                    # (A.T @ B).T, contiguous
            elif cfg.grad_weight:
                grad_weight = _dynamic_mm(x2d.T, go2d, sr, cfg.dtype).T
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
