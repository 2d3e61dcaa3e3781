"""Tensor subclasses of the quantized-training hot path on MI355X.

In scope (reference subclasses/__init__.py names): the INT8 quantizer, `Int8LinearWeight`, the INT8
mixed-precision weight and the BitNet training weight (the scalar-column-scale caller of `scaled_mm`,
SURVEY.md section 8f rank 4).  The INT4 weight-only subclass is outside the scaled INT8 / FP8 matmul
path (no INT4 MFMA on CDNA4); its names import and raise when used.
"""
from .bitnet import BitNetTrainingLinearWeight, convert_bitnet
from .int8 import Int8LinearWeight, Int8QTConfig, convert_int8_quantized_training, quantize_int8
from .mixed_precision import MixedPrecisionConfig, MixedPrecisionLinearWeight, convert_mixed_precision


def _out_of_scope(name: str):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"subclasses.{name} is outside the MI355X INT8/FP8 scaled-matmul hot path")

    fn.__name__ = name
    return fn


Int4LinearWeight = _out_of_scope("Int4LinearWeight")
convert_int4_quantized_training = _out_of_scope("convert_int4_quantized_training")

__all__ = [
    "BitNetTrainingLinearWeight",
    "convert_bitnet",
    "Int4LinearWeight",
    "convert_int4_quantized_training",
    "Int8LinearWeight",
    "Int8QTConfig",
    "convert_int8_quantized_training",
    "MixedPrecisionConfig",
    "MixedPrecisionLinearWeight",
    "convert_mixed_precision",
    "quantize_int8",
]
