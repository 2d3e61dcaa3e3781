"""Tile-scaled GEMM at 4096^3, the bench's cases: DeepSeek-style (1 x 128, 128 x 128) and Jetfire-style 32^3 blocks, int8 and fp8, three
layouts; also the row/col-scaled 128 x 128 kernel.  One line per case (median of 20 event-timed launches): for A/B runs of library builds."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi
from tools.cmp_variants import lay, timeit

S = 4096
torch.manual_seed(0)
tsA = torch.randn(S, S // 128, device="cuda").div(128).bfloat16()
tsB = torch.randn(S // 128, S // 128, device="cuda").div(128).bfloat16()
jsA = torch.randn(S // 32, S // 32, device="cuda").div(128).bfloat16()
jsB = torch.randn(S // 32, S // 32, device="cuda").div(128).bfloat16()
rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
L = _cabi.load()
for kind in ("i8", "f8"):
    for name, (acm, bcm) in {"nt": (False, True), "nn": (False, False), "tn": (True, False)}.items():
        if kind == "i8":
            A = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), acm)
            B = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), bcm)
        else:
            A = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), acm)
            B = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), bcm)
        td = timeit(lambda: gn_kernels.scaled_mm(A, B, lay(tsA, acm), lay(tsB, bcm)), iters=20)
        tj = timeit(lambda: gn_kernels.scaled_mm(A, B, jsA, jsB), iters=10)
        L.gnk_set_gemm_variant(1)
        t1 = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs), iters=20)
        L.gnk_set_gemm_variant(-1)
        f = lambda t: f"{t:7.1f}us {2 * S ** 3 / t / 1e6:6.0f}T"
        print(f"{kind} {name}: deepseek {f(td)} | jetfire32 {f(tj)} | rowcol 128x128 {f(t1)}", flush=True)
