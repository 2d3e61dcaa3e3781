"""[needs an ABLATE build: make -C quantized-training_amd/csrc ABLATE=1 OUT=../gn_kernels/libgnk_hip_ablate.so BUILD=build_ablate, then GNK_LIBRARY=.../libgnk_hip_ablate.so] MXFP4 GEMM and quantize_mx timings (event-timed), FP4 rate against the 10 PFLOP/s dense peak."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

from gn_kernels import mx
from tools.cmp_variants import timeit

for S in (1024, 2048, 4096, 8192):
    A = torch.randint(0, 256, (S, S // 2), dtype=torch.uint8, device="cuda").view(mx.FP4)
    B = torch.randint(0, 256, (S, S // 2), dtype=torch.uint8, device="cuda").view(mx.FP4).T
    sA = torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device="cuda").view(mx.E8M0)
    sB = torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device="cuda").view(mx.E8M0)
    t = timeit(lambda: mx.mxfp4_mm(A, B, sA, sB))
    from gn_kernels import _cabi
    res = []
    for flags in (128, 1024, 128 + 1024, 256):
        _cabi.load().gnk_set_debug_flags(flags)
        res.append(f"flags {flags}: {timeit(lambda: mx.mxfp4_mm(A, B, sA, sB)):7.1f} us")
    _cabi.load().gnk_set_debug_flags(0)
    print("   ablations (128 = no scale DMA, 1024 = no scale LDS reads, 256 = 128^2 kernel): " + " | ".join(res))
    print(f"mxfp4_mm {S}^3: {t:8.1f} us  {2 * S ** 3 / t / 1e6:7.0f} TFLOP/s  ({2 * S ** 3 / t / 1e6 / 10066 * 100:4.1f} % of 10.07 PF)", flush=True)
for shape in ((4096, 4096), (16384, 8192)):
    x = torch.randn(shape, device="cuda").bfloat16()
    for name, dt in (("fp4", mx.FP4), ("e4m3", torch.float8_e4m3fn)):
        t = timeit(lambda: mx.quantize_mx(x, dt))
        out_b = 0.5 if name == "fp4" else 1.0
        bytes_ = x.numel() * (2 + out_b + 1 / 32)
        print(f"quantize_mx {shape} bf16 -> {name}: {t:8.1f} us  {bytes_ / t / 1e3:7.1f} GB/s", flush=True)
