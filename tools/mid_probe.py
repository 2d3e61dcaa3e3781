"""Mid-size cubes (benchmark_mm.py's 1024^3 / 2048^3) per kernel variant: event-timed through the op and HIP-graph replayed (kernel time)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from bench import time_kernel, time_kernel_graph
from gn_kernels import _cabi
from tools.cmp_variants import lay

L = _cabi.load()
for S in (1024, 2048, 3072):
    rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
    cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
    for kind in ("i8", "f8"):
        for name, (acm, bcm) in {"nt": (False, True), "nn": (False, False), "tn": (True, False)}.items():
            if kind == "i8":
                A = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), acm)
                B = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), bcm)
            else:
                A = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), acm)
                B = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), bcm)
            row = []
            for vn, v in (("auto", -1), ("128", 1), ("128deep", 5), ("256", 0)):
                L.gnk_set_gemm_variant(v)
                fn = lambda: gn_kernels.scaled_mm(A, B, rs, cs)
                ev, _ = time_kernel(fn, iters=30)
                gr = time_kernel_graph(fn)
                row.append(f"{vn} {ev * 1e3:5.1f}/{gr * 1e3:5.1f}us {2 * S ** 3 / gr / 1e9:5.0f}T")
            L.gnk_set_gemm_variant(-1)
            print(f"{S}^3 {kind} {name}: " + " | ".join(row) + "   (event / graph-replay us, TOPS of the replay)", flush=True)
