"""The fallback GEMM (gnk_set_gemm_variant 2) at 2048^3 in the three layouts, and a tile-scaled case with QK = 16 (always on that kernel)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


S = 2048
rs = torch.rand(S, 1, device="cuda").bfloat16()
cs = torch.rand(1, S, device="cuda").bfloat16()
for name, (acm, bcm) in {"A@B.T": (False, True), "A@B": (False, False), "A.T@B": (True, False)}.items():
    A = torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda")
    B = torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda")
    A = A.T.contiguous().T if acm else A
    B = B.T.contiguous().T if bcm else B
    L.gnk_set_gemm_variant(2)
    t = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs))
    L.gnk_set_gemm_variant(-1)
    print(f"generic {name:6s} {S}^3: {t:8.1f} us  {2 * S ** 3 / t / 1e6:6.1f} TOP/s", flush=True)
    sA = torch.rand(S // 4, S // 16, device="cuda").bfloat16()
    sB = torch.rand(S // 16, S // 8, device="cuda").bfloat16()
    t = timeit(lambda: gn_kernels.scaled_mm(A, B, sA, sB))
    print(f"tile-scaled (4, 8, 16) blocks {name:6s} {S}^3: {t:8.1f} us  {2 * S ** 3 / t / 1e6:6.1f} TOP/s", flush=True)
