"""[needs an ABLATE build: make -C quantized-training_amd/csrc ABLATE=1 OUT=../gn_kernels/libgnk_hip_ablate.so BUILD=build_ablate, then GNK_LIBRARY=.../libgnk_hip_ablate.so] Row-wise quantize 4096x4096 bf16 under the experiment flags (run under rocprofv3 for kernel times)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels  # noqa: F401
from gn_kernels import _cabi
from subclasses import quantize_int8

L = _cabi.load()
x = torch.randn(4096, 4096, device="cuda").bfloat16()
for flags in (0, 8, 0, 8):
    L.gnk_set_debug_flags(flags)
    for _ in range(20):
        quantize_int8(x, dim=1)
    torch.cuda.synchronize()
L.gnk_set_debug_flags(0)
