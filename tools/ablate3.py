"""[needs an ABLATE build: make -C quantized-training_amd/csrc ABLATE=1 OUT=../gn_kernels/libgnk_hip_ablate.so BUILD=build_ablate, then GNK_LIBRARY=.../libgnk_hip_ablate.so] Fixed cost of the 8-phase 256x256 kernel (variant 4) at K = 256: kernel durations under rocprofv3 for the ablation flags."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()
S = 4096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda")
Bt = torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda")
rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
L.gnk_set_gemm_variant(4)
for flags in (0, 8, 7, 15):
    L.gnk_set_debug_flags(flags)
    for _ in range(20):
        gn_kernels.scaled_mm(A, Bt.T, rs, cs)
    torch.cuda.synchronize()
    print("flags", flags, "done")
L.gnk_set_debug_flags(0)
