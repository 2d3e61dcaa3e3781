"""Model-conversion glue kept under the reference's module and function name (train_utils.py:35-75): the training scripts only call
`quantize_model(model, quantize, **kwargs)` with `--quantize {mixed_precision,int8_quantized_training,bitnet}` and JSON kwargs
(llm_pretrain.py:45,122).  Nothing else of the reference's train_utils (optimizer factory over bitsandbytes / torchao, LR schedule)
belongs to the scaled-matmul path."""
from __future__ import annotations

from torch import nn

from subclasses import (Int8QTConfig, MixedPrecisionConfig, convert_bitnet, convert_int8_quantized_training,
                        convert_mixed_precision)


def quantize_model(model: nn.Module, quantize: str | None, **kwargs):
    if quantize == "mixed_precision":
        config = MixedPrecisionConfig(**kwargs)
        print(f"Mixed precision with {config=}")
        convert_mixed_precision(model, config=config)
    elif quantize == "int8_quantized_training":
        config = Int8QTConfig(**kwargs)
        print(f"INT8 quantized training with {config=}")
        convert_int8_quantized_training(model, config=config)
    elif quantize == "bitnet":
        # the reference also re-plumbs LlamaDecoderLayer's RMSNorms in front of every Linear (train_utils.py:48-71): model surgery,
        # outside this path -- convert the Linears and leave the norms to the caller
        convert_bitnet(model, **kwargs)
    else:
        assert quantize is None
    return model
