"""[needs an ABLATE build: make -C quantized-training_amd/csrc ABLATE=1 OUT=../gn_kernels/libgnk_hip_ablate.so BUILD=build_ablate, then GNK_LIBRARY=.../libgnk_hip_ablate.so] Fixed-overhead and per-phase cost of the 256x256 ping-pong GEMM (GPU box): time vs K at M=N=4096."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

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


S = 4096
rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
L.gnk_set_gemm_variant(4)   # the 8-phase kernel is the one that carries the ablation flags
for K in (128, 256, 1024, 4096, 8192, 16384):
    A = torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda")
    Bt = torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda")
    row = []
    for flags in (0, 7, 5, 6):
        L.gnk_set_debug_flags(flags)
        row.append(timeit(lambda: gn_kernels.scaled_mm(A, Bt.T, rs, cs)))
    L.gnk_set_debug_flags(0)
    i32 = timeit(lambda: gn_kernels.int8_mm(A, Bt.T))
    print(f"K={K:6d}: full {row[0]:7.1f} us | barriers only {row[1]:7.1f} | mfma+barriers {row[2]:7.1f} | dma+barriers {row[3]:7.1f} | int32-out full {i32:7.1f}", flush=True)
