"""Every GEMM of one "Llama2-1B" layer's Linears at the training shapes (16 x 2048 tokens), as subclasses/mixed_precision.py issues them:
forward A@B.T, grad-input A@B, grad-weight (A.T@B).T through scaled_mm_t; per kernel variant (gnk_set_gemm_variant) and the heuristic."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()
T = 16 * 2048


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


def i8(*shape):
    return torch.randint(-128, 128, shape, dtype=torch.int8, device="cuda")


def sc(*shape):
    return (torch.rand(*shape, device="cuda") * 0.01 + 1e-3).bfloat16()


tot = {}
for name, (K, N), count in (("q/k/v/o", (2048, 2048), 4), ("gate/up", (2048, 5632), 2), ("down", (5632, 2048), 1)):
    x, w, g = i8(T, K), i8(N, K), i8(T, N)
    sT, sN, sK, sK1 = sc(T, 1), sc(1, N), sc(1, K), sc(K, 1)
    cases = {
        "fwd   x @ w.T      ": lambda: gn_kernels.scaled_mm(x, w.T, sT, sN),
        "dgrad g @ w        ": lambda: gn_kernels.scaled_mm(g, w, sT, sK),
        "wgrad (x.T @ g).T  ": lambda: gn_kernels.scaled_mm_t(x.T, g, sK1, sN),
    }
    for cname, fn in cases.items():
        row = []
        for vn, v in (("pp8", 4), ("w4", 5), ("auto", -1), ("auto-no-w4", -2), ("pp2", 0), ("auto", -1), ("auto-no-w4", -2)):
            L.gnk_set_gemm_variant(v)
            try:
                t = timeit(fn)
                row.append(f"{vn} {t:6.1f} {2 * T * K * N / t / 1e6:4.0f}T")
                if vn.startswith("auto"):
                    tot[vn + " " + cname] = tot.get(vn + " " + cname, 0.0) + count * t / 2
            except Exception as exc:
                row.append(f"{vn} failed")
        L.gnk_set_gemm_variant(-1)
        print(f"{name:8s} K={K:5d} N={N:5d} {cname}: " + " | ".join(row), flush=True)
print({k: f"{v * 22 / 1e3:.1f} ms per step" for k, v in tot.items()})
