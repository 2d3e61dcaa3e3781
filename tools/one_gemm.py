"""One GEMM configuration in a loop, for rocprofv3 (kernel trace or PMC passes).
usage: python3 tools/one_gemm.py [i8|f8] [nt|nn|tn] [K] [iters] [variant] [tile block: 0 = row/col scales, 128 = (1, 128, 128) blocks, 32 / 64 = square blocks]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi

kind = sys.argv[1] if len(sys.argv) > 1 else "i8"
layout = sys.argv[2] if len(sys.argv) > 2 else "nt"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
variant = int(sys.argv[5]) if len(sys.argv) > 5 else -1
S = 4096
acm, bcm = {"nt": (False, True), "nn": (False, False), "tn": (True, False)}[layout]
lay = lambda x, cm: x.T.contiguous().T if cm else x.contiguous()
if kind == "i8":
    A = lay(torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda"), acm)
    B = lay(torch.randint(-128, 127, (K, S), dtype=torch.int8, device="cuda"), bcm)
else:
    A = lay(torch.randn(S, K, device="cuda").to(torch.float8_e4m3fn), acm)
    B = lay(torch.randn(K, S, device="cuda").to(torch.float8_e4m3fn), bcm)
rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
blk = int(sys.argv[6]) if len(sys.argv) > 6 else 0
if blk:
    rs = torch.randn(S if blk == 128 else S // blk, K // blk, device="cuda").div(128).bfloat16()
    cs = torch.randn(K // blk, S // blk, device="cuda").div(128).bfloat16()
_cabi.load().gnk_set_gemm_variant(variant)
import os
if os.environ.get("GNK_TILE_RESCALE"):      # exact | fused | auto (gn_kernels.mm.set_tile_rescale)
    from gn_kernels import mm
    mm.set_tile_rescale(os.environ["GNK_TILE_RESCALE"])
for _ in range(iters):
    gn_kernels.scaled_mm(A, B, rs, cs)
torch.cuda.synchronize()
print("done", kind, layout, K, iters, variant)
