"""Tile-scaled int8/fp8 at 4096^3: operand layout x scale layout, event-timed."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from tools.cmp_variants import lay, timeit

S = 4096
tsA = torch.randn(S, S // 128, device="cuda").div(128).bfloat16()
tsB = torch.randn(S // 128, S // 128, device="cuda").div(128).bfloat16()
for kind in ("i8", "f8"):
    for name, (acm, bcm) in {"nt": (False, True), "nn": (False, False), "tn": (True, False), "tt": (True, True)}.items():
        if kind == "i8":
            A = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), acm)
            B = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), bcm)
        else:
            A = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), acm)
            B = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), bcm)
        row = []
        for sa_cm in (False, True):
            for sb_cm in (False, True):
                a, b = lay(tsA, sa_cm), lay(tsB, sb_cm)
                t = timeit(lambda: gn_kernels.scaled_mm(A, B, a, b), iters=20)
                row.append(f"sA {'col' if sa_cm else 'row'} sB {'col' if sb_cm else 'row'}: {t:6.1f}us")
        print(kind, name, " | ".join(row), flush=True)
