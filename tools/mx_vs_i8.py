"""[needs an ABLATE build: make -C quantized-training_amd/csrc ABLATE=1 OUT=../gn_kernels/libgnk_hip_ablate.so BUILD=build_ablate, then GNK_LIBRARY=.../libgnk_hip_ablate.so] Same bytes, same tile program: int8 4-stage kernel at K = 2048 bytes vs the 256 x 256 MXFP4 kernel at K = 4096 (2048 bytes), random and zero data."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi, mx
from tools.cmp_variants import timeit

L = _cabi.load()
S = 4096
for zero in (False, True):
    A8 = torch.zeros(S, S // 2, dtype=torch.int8, device="cuda") if zero else torch.randint(-128, 127, (S, S // 2), dtype=torch.int8, device="cuda")
    B8 = (torch.zeros(S, S // 2, dtype=torch.int8, device="cuda") if zero else torch.randint(-128, 127, (S, S // 2), dtype=torch.int8, device="cuda")).T
    rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
    cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
    L.gnk_set_gemm_variant(0)
    t_i8 = timeit(lambda: gn_kernels.scaled_mm(A8, B8, rs, cs))
    t_f8 = timeit(lambda: gn_kernels.scaled_mm(A8.view(torch.float8_e4m3fn) if zero else A8.view(torch.uint8).clamp(0, 119).view(torch.float8_e4m3fn), (B8.T.contiguous().view(torch.uint8).clamp(0, 119).view(torch.float8_e4m3fn)).T, rs, cs))
    L.gnk_set_gemm_variant(-1)
    A4 = A8.view(torch.uint8).view(mx.FP4)
    B4 = B8.T.contiguous().view(torch.uint8).view(mx.FP4).T
    sA = torch.full((S, S // 32), 127, dtype=torch.uint8, device="cuda") if zero else torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device="cuda")
    sB = sA.clone()
    out = []
    for flags in (512, 512 + 128 + 1024):
        L.gnk_set_debug_flags(flags)
        out.append(f"flags {flags}: {timeit(lambda: mx.mxfp4_mm(A4, B4, sA.view(mx.E8M0), sB.view(mx.E8M0))):6.1f} us")
    L.gnk_set_debug_flags(0)
    print(f"{'zeros ' if zero else 'random'}: int8 pp2 4096x4096x2048B {t_i8:6.1f} us | fp8 pp2 {t_f8:6.1f} us | mxfp4 256^2 " + " | ".join(out), flush=True)
