"""Training-shape GEMMs (Llama-470m / 1B, T = 8192 / 32768 tokens): auto plan vs forced 256^2 / 128^2, event-timed."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi
from tools.cmp_variants import lay, timeit

L = _cabi.load()
shapes = []
for T, h, f in ((8192, 1024, 4096), (32768, 2048, 5632)):
    shapes += [(T, h, h, "nt"), (T, f, h, "nt"), (T, h, f, "nt"),          # forward   x @ W.T
               (T, h, h, "nn"), (T, h, f, "nn"), (T, f, h, "nn")]          # grad-input dY @ W
for M, N, K, name in shapes:
    acm, bcm = {"nt": (False, True), "nn": (False, False), "tn": (True, False)}[name]
    A = lay(torch.randint(-128, 127, (M, K), dtype=torch.int8, device="cuda"), acm)
    B = lay(torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda"), bcm)
    rs = torch.randn(M, 1, device="cuda").div(128).bfloat16()
    cs = torch.randn(1, N, device="cuda").div(128).bfloat16()
    row = []
    for v in (-1, 0, 1):
        L.gnk_set_gemm_variant(v)
        t = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs), iters=20)
        row.append(f"{'auto' if v < 0 else ('256^2' if v == 0 else '128^2')} {t:7.1f}us {2 * M * N * K / t / 1e6:6.0f}T")
    L.gnk_set_gemm_variant(-1)
    print(f"{M}x{N}x{K} {name}: " + " | ".join(row), flush=True)
