"""pp2 (4-stage) vs pp8 (8-phase) for A@B (B N-major) at training shapes, alternating, 3 rounds."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(s, e))
    return ts[len(ts) // 2] * 1e3


for layout in ("A@B", "A@B.T"):
    for (M, N, K) in [(32768, 2048, 5632), (32768, 5632, 2048), (32768, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (32768, 2048, 8192), (16384, 4096, 5632)]:
        A = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda")
        B = torch.randint(-128, 128, (K, N), dtype=torch.int8, device="cuda") if layout == "A@B" else torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda").T
        sa = torch.rand(M, 1, device="cuda").bfloat16()
        sb = torch.rand(1, N, device="cuda").bfloat16()
        res = {0: [], 4: []}
        for _ in range(3):
            for v in (0, 4):
                L.gnk_set_gemm_variant(v)
                res[v].append(timeit(lambda: gn_kernels.scaled_mm(A, B, sa, sb)))
        L.gnk_set_gemm_variant(-1)
        print(f"{layout:6s} {M:6d} x {N:5d} x {K:5d}: pp2 {min(res[0]):7.1f} us ({', '.join(f'{t:.0f}' for t in res[0])}) | pp8 {min(res[4]):7.1f} us ({', '.join(f'{t:.0f}' for t in res[4])}) | pp2/pp8 {min(res[0]) / min(res[4]):.3f}", flush=True)
        del A, B
