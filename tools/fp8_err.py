"""Diagnostic: accumulation error of the fp8 GEMM kernels vs an fp64 reference (run on the GPU box)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

torch.manual_seed(0)
L = _cabi.load()
for K in (128, 512, 4096):
    M = N = 256
    A = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
    B = torch.randn(N, K, device="cuda").to(torch.float8_e4m3fn).T
    ref = A.double() @ B.double()
    mag = A.double().abs() @ B.double().abs()
    for name, v in (("tile128 (f8f6f4 16x16x128)", 1), ("generic (fp8 16x16x32)", 2)):
        L.gnk_set_gemm_variant(v)
        out = gn_kernels._triton_mm(A, B, out_dtype=torch.float32).double()
        err = (out - ref).abs()
        print(f"K={K} {name}: max|err|/sum|prod| = {(err / mag).max().item():.3e}, max|err| = {err.max().item():.3e}, "
              f"max|err|/max|ref| = {(err.max() / ref.abs().max()).item():.3e}, bf16-default-close: "
              f"{torch.allclose(out.float().bfloat16().float(), ref.float().bfloat16().float(), rtol=1.6e-2, atol=1e-5)}")
    L.gnk_set_gemm_variant(-1)
    out = torch.mm(A.float(), B.float()).double()
    err = (out - ref).abs()
    print(f"K={K} torch fp32 mm: max|err|/sum|prod| = {(err / mag).max().item():.3e}")
