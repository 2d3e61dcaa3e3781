"""MXFP8 GEMM: 128 x 128 kernel against the 256 x 256 8-phase kernel (gnk_set_gemm_variant 1 / 4), HIP-graph-free event timing."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

from gn_kernels import _cabi, mx

L = _cabi.load()


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(s, e))
    return ts[len(ts) // 2] * 1e3


for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 4096, 1024), (2048, 2048, 2048), (32768, 2048, 2048)]:
    A = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
    B = torch.randn(N, K, device="cuda").to(torch.float8_e4m3fn).T
    sa = torch.randint(120, 134, (M, K // 32), dtype=torch.uint8, device="cuda").view(mx.E8M0)
    sb = torch.randint(120, 134, (N, K // 32), dtype=torch.uint8, device="cuda").view(mx.E8M0)
    out = []
    for name, v in (("128", 1), ("256", 4), ("auto", -1)):
        L.gnk_set_gemm_variant(v)
        t = timeit(lambda: mx.mxfp8_mm(A, B, sa, sb))
        out.append(f"{name} {t:7.1f} us {2 * M * N * K / t / 1e6:6.0f} TFLOP/s")
    L.gnk_set_gemm_variant(-1)
    print(f"{M} x {N} x {K}: " + " | ".join(out), flush=True)
