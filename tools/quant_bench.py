"""Quantize-pass loop for rocprofv3 --kernel-trace --stats (kernel time without host overhead) + event timing."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels  # noqa: F401
from subclasses import quantize_int8
from tools.cmp_variants import timeit

for R, C in ((4096, 4096), (8192, 1024), (8192, 4096), (16384, 8192)):
    x = torch.randn(R, C, device="cuda").bfloat16()
    nbytes = 3 * R * C + 2 * R
    for dim in (1, 0):
        t = timeit(lambda: quantize_int8(x, dim=dim), iters=30)
        print(f"{R}x{C} dim={dim}: {t:7.1f} us (event-timed, incl. host overhead)  {nbytes / t * 1e-6:7.1f} GB/s", flush=True)
