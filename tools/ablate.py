"""[needs an ABLATE build: make -C quantized-training_amd/csrc ABLATE=1 OUT=../gn_kernels/libgnk_hip_ablate.so BUILD=build_ablate, then GNK_LIBRARY=.../libgnk_hip_ablate.so] Timing ablation of the 256x256 ping-pong GEMM (run on the GPU box): which of LDS-DMA, LDS reads and
MFMA bounds the K loop.  Results of ablated runs are wrong by construction; only times matter."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()
L.gnk_set_debug_flags.argtypes = [__import__("ctypes").c_int]


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


S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
A = torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda")
Bt = torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda")
A8 = torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn)
B8t = torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn)
rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
names = {0: "full", 1: "no DMA", 2: "no MFMA", 4: "no LDS reads", 3: "no DMA, no MFMA", 5: "no DMA, no reads (MFMA + barriers)",
         6: "no MFMA, no reads (DMA + barriers)", 7: "barriers only"}
for tag, (a, b) in {"int8 NT": (A, Bt.T), "int8 NN": (A, Bt), "int8 TN": (A.T, Bt), "fp8 NT": (A8, B8t.T)}.items():
    for v in (4, 3):   # 4 = the 8-phase kernel, the only one that carries the ablation flags (int8 only); 3 = first kernel
        if not tag.startswith("int8"):
            continue
        L.gnk_set_gemm_variant(v)
        for flags, nm in names.items():
            if v == 3 and flags:
                continue
            L.gnk_set_debug_flags(flags)
            us = timeit(lambda: gn_kernels.scaled_mm(a, b, rs, cs))
            print(f"{tag} variant {v} {nm:38s}: {us:8.1f} us  {2 * S ** 3 / us * 1e-6:8.1f} TOPS", flush=True)
    L.gnk_set_debug_flags(0)
    L.gnk_set_gemm_variant(-1)
