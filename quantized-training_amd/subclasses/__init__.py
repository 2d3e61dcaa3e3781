"""Tensor subclasses of the quantized-training hot path on MI355X.

In scope (reference subclasses/__init__.py names): the INT8 quantizer, `Int8LinearWeight` and the
INT8 mixed-precision weight.  BitNet and INT4 weight-only subclasses are outside the scaled INT8 /
FP8 matmul path; their names import and raise when used.
"""
from .int8 import Int8LinearWeight, Int8QTConfig, convert_int8_quantized_training, quantize_int8
from .mixed_precision import MixedPrecisionConfig, MixedPrecisionLinearWeight, convert_mixed_precision


def _out_of_scope(name: str):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"subclasses.{name} is outside the MI355X INT8/FP8 scaled-matmul hot path")

    fn.__name__ = name
    return fn


BitNetTrainingLinearWeight = _out_of_scope("BitNetTrainingLinearWeight")
convert_bitnet = _out_of_scope("convert_bitnet")
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
