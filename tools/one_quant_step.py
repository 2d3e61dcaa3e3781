"""A few launches of the in-step quantize kernels on cold tensors for rocprofv3 PMC passes: tools/one_quant_step.py <R> <C>"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

from gn_kernels import quantize as qmod

R, C = int(sys.argv[1]), int(sys.argv[2])
xs = [torch.randn(R, C, device="cuda").bfloat16() for _ in range(3)]
cms = [qmod.quantize_int8_rowwise_colmax(x)[2] for x in xs]
for it in range(12):
    x, cm = xs[it % 3], cms[it % 3]
    qmod.quantize_int8(x, dim=1)
    qmod.quantize_int8_rowwise_colmax(x)
    qmod.quantize_int8_colwise_given(x, cm)
torch.cuda.synchronize()
