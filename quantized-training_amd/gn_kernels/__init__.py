"""gn_kernels for MI355X (gfx950): the reference's operator package, re-implemented on hand-written
HIP kernels behind a C ABI (include/gnk.h).  Public names follow kernels/gn_kernels/__init__.py of
the reference; names outside the INT8/FP8 scaled-matmul hot path and its MX extension import fine and raise
NotImplementedError when called."""
from . import _cabi
from . import bitnet  # noqa: F401  (registers the bitnet_* device ops)
from .fp8 import fp8_mm, scaled_fp8_mm
from .mm import _triton_mm, int8_mm, scaled_mm, scaled_mm_t
from .mx import dequantize_mxfp4, mxfp4_mm, mxfp8_mm, pack_block_scales_nv, quantize_mx
from .quantize import quantize_int8


def _out_of_scope(name: str, why: str):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"gn_kernels.{name} is outside the MI355X INT8/FP8 scaled-matmul hot path ({why})")

    fn.__name__ = name
    return fn


int4_mm = _out_of_scope("int4_mm", "no INT4 MFMA on CDNA4")
scaled_int4_mm = _out_of_scope("scaled_int4_mm", "no INT4 MFMA on CDNA4")
nvfp4_mm = _out_of_scope("nvfp4_mm", "NVIDIA-specific format")
_triton_conv2d = _out_of_scope("_triton_conv2d", "conv2d is unmaintained upstream and not on the path")
int8_conv2d = _out_of_scope("int8_conv2d", "conv2d is unmaintained upstream and not on the path")
scaled_int8_conv2d = _out_of_scope("scaled_int8_conv2d", "conv2d is unmaintained upstream and not on the path")
quantize_nvfp4 = _out_of_scope("quantize_nvfp4", "NVIDIA-specific format")
quantize_nvfp4_triton = _out_of_scope("quantize_nvfp4_triton", "NVIDIA-specific format")

__all__ = [
    "int4_mm",
    "scaled_int4_mm",
    "scaled_fp8_mm",
    "fp8_mm",
    "mxfp4_mm",
    "mxfp8_mm",
    "nvfp4_mm",
    "_triton_conv2d",
    "int8_conv2d",
    "scaled_int8_conv2d",
    "_triton_mm",
    "int8_mm",
    "scaled_mm",
    "scaled_mm_t",
    "quantize_int8",
    "quantize_mx",
    "quantize_nvfp4",
    "quantize_nvfp4_triton",
    "dequantize_mxfp4",
    "pack_block_scales_nv",
]
