"""Tile-scaled mm with Jetfire-style 32 x 32 blocks (BASELINE configs[2]) at 2048 / 4096: event-timed, checked against the fp32 formula."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from tools.cmp_variants import lay, timeit

for S in (2048, 4096):
    for Q in (32, 64, 128):
        sA = torch.randn(S // Q, S // Q, device="cuda").div(128).bfloat16()
        sB = torch.randn(S // Q, S // Q, device="cuda").div(128).bfloat16()
        for kind in ("i8", "f8"):
            for name, (acm, bcm) in {"nt": (False, True), "nn": (False, False), "tn": (True, False)}.items():
                if kind == "i8":
                    A = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), acm)
                    B = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), bcm)
                else:
                    A = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), acm)
                    B = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), bcm)
                t = timeit(lambda: gn_kernels.scaled_mm(A, B, sA, sB), iters=5)
                print(f"{S} blocks {Q}x{Q}x{Q} {kind} {name}: {t:9.1f} us  {2 * S ** 3 / t / 1e6:7.1f} TOP/s", flush=True)
