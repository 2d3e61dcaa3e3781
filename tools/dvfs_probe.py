"""Does operand data (toggle rate -> power -> clock) move GEMM time?  Same kernels, different data."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi
from tools.cmp_variants import timeit

L = _cabi.load()
S, K = 4096, 8192
rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
data = {
    "zeros": lambda: torch.zeros(S, K, dtype=torch.int8, device="cuda"),
    "pm1": lambda: torch.randint(-1, 2, (S, K), dtype=torch.int8, device="cuda"),
    "randn*16": lambda: torch.randn(S, K, device="cuda").mul(16).round().clamp(-128, 127).to(torch.int8),
    "uniform": lambda: torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda"),
}
for name, mk in data.items():
    A, Bt = mk(), mk()
    row = []
    for vn, v in {"pp2": 0, "pp8": 4}.items():
        L.gnk_set_gemm_variant(v)
        t = timeit(lambda: gn_kernels.scaled_mm(A, Bt.T, rs, cs))
        row.append(f"i8 {vn} {t:7.1f}us")
    L.gnk_set_gemm_variant(0)
    for fmt in (torch.float8_e4m3fn,):
        A8, B8 = A.view(fmt), Bt.view(fmt)       # same BYTES reinterpreted (may hold NaNs: timing only)
        t = timeit(lambda: gn_kernels.scaled_mm(A8, B8.T, rs, cs))
        row.append(f"f8(same bytes) pp2 {t:7.1f}us")
    L.gnk_set_gemm_variant(-1)
    print(f"{name:9s}: " + " | ".join(row), flush=True)
