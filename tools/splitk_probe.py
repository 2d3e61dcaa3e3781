"""Split-K on/off (forced variant 1 never splits) for grad-weight-like shapes, event-timed."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi
from tools.cmp_variants import lay, timeit

L = _cabi.load()
for M, N, K, name in ((1024, 1024, 8192, "tn"), (1024, 4096, 8192, "tn"), (4096, 1024, 8192, "tn"), (2048, 2048, 2048, "nt"), (2048, 2048, 8192, "tn"),
                      (2048, 5632, 32768, "tn"), (2048, 2048, 32768, "tn"), (1024, 1024, 2048, "nt"), (512, 512, 4096, "nt")):
    acm, bcm = {"nt": (False, True), "nn": (False, False), "tn": (True, False)}[name]
    A = lay(torch.randint(-128, 127, (M, K), dtype=torch.int8, device="cuda"), acm)
    B = lay(torch.randint(-128, 127, (K, N), dtype=torch.int8, device="cuda"), bcm)
    rs = torch.randn(M, 1, device="cuda").div(128).bfloat16()
    cs = torch.randn(1, N, device="cuda").div(128).bfloat16()
    ws = L.gnk_scaled_mm_workspace_bytes(_cabi.I8, _cabi.I8, M, N, K)
    t_auto = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs))
    L.gnk_set_gemm_variant(1)
    t_128 = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs))
    L.gnk_set_gemm_variant(0)
    try:
        t_256 = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs))
    except RuntimeError:
        t_256 = float("nan")
    L.gnk_set_gemm_variant(-1)
    print(f"{M}x{N}x{K} {name}: splits {ws // (M * N * 4) if ws else 1}  auto {t_auto:7.1f} us | 128^2 unsplit {t_128:7.1f} us | 256^2 {t_256:7.1f} us", flush=True)
