"""A few quantize launches of one shape for the PMC passes: tools/one_quant.py <R> <C> <dim>"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels  # noqa: F401
from subclasses import quantize_int8

R, C, dim = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(R, C, device="cuda").bfloat16()
for _ in range(12):
    quantize_int8(x, dim=dim)
torch.cuda.synchronize()
