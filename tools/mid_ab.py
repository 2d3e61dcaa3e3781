"""The 128 x 128 kernel on mid-size and Llama-470m shapes (heuristic choice and forced variant 1), region-timed after a pre-warm: for A/B library
builds through GNK_LIBRARY (tools/ab_libs.sh)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import bench
import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()
lay = lambda x, cm: x.T.contiguous().T if cm else x.contiguous()
LAY = {"nt": (False, True), "nn": (False, False), "tn": (True, False)}
shapes = [(1024, 1024, 1024, "nt"), (1024, 1024, 1024, "tn"), (2048, 2048, 2048, "nt"), (2048, 2048, 2048, "nn"), (2048, 2048, 2048, "tn"),
          (8192, 1024, 1024, "nt"), (8192, 1024, 4096, "nt"), (8192, 1024, 1024, "nn"), (8192, 1024, 4096, "nn"),
          (1024, 1024, 8192, "tn"), (4096, 1024, 8192, "tn"), (1024, 4096, 8192, "tn"), (4096, 4096, 4096, "nt")]
for M, N, K, name in shapes:
    acm, bcm = LAY[name]
    A = lay(torch.randint(-128, 127, (M, K), dtype=torch.int8, device="cuda"), acm)
    B = lay(torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda"), bcm)
    rs = torch.randn(M, 1, device="cuda").div(128).bfloat16()
    cs = torch.randn(1, N, device="cuda").div(128).bfloat16()
    out = []
    for vn, v in (("128x128", 1), ("64x128", 6), ("auto", -1)):
        L.gnk_set_gemm_variant(v)
        t, _ = bench.time_kernel(lambda: gn_kernels.scaled_mm(A, B, rs, cs), iters=60)
        out.append(f"{vn} {t * 1e3:6.1f} us {2 * M * N * K / t * 1e-9:5.0f} T")
    L.gnk_set_gemm_variant(-1)
    print(f"{M:5d}x{N:5d}x{K:5d} {name}: " + " | ".join(out), flush=True)
