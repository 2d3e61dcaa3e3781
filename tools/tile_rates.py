"""Tile-scaled GEMM rates at 4096^3 (DeepSeek-style 1x128 / 128x128 blocks and Jetfire-style 32^3 blocks, bf16 scales, default rescale mode),
region-timed after a pre-warm: TOP/s per operand layout."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import bench
import gn_kernels

S = 4096
import os
if os.environ.get("GNK_VARIANT"):       # force a tile variant (csrc/gnk_gemm_dispatch.h), e.g. 6 = 64 x 128 tiles
    from gn_kernels import _cabi
    _cabi.load().gnk_set_gemm_variant(int(os.environ["GNK_VARIANT"]))
lay = lambda x, cm: x.T.contiguous().T if cm else x.contiguous()
for kind in ("i8", "f8"):
    for blk in (128, 32):
        out = []
        for name, (acm, bcm) in {"A@B.T": (False, True), "A@B": (False, False), "A.T@B": (True, False)}.items():
            if kind == "i8":
                A = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), acm)
                B = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), bcm)
            else:
                A = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), acm)
                B = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), bcm)
            sA = torch.randn(S if blk == 128 else S // blk, S // blk, device="cuda").div(128).bfloat16()
            sB = torch.randn(S // blk, S // blk, device="cuda").div(128).bfloat16()
            avg, _ = bench.time_kernel(lambda: gn_kernels.scaled_mm(A, B, sA, sB), iters=40)
            out.append(f"{name} {avg * 1e3:6.1f} us {2 * S ** 3 / avg * 1e-9:5.0f} T")
        print(f"{kind} blocks {'1x128/128x128' if blk == 128 else '32x32x32'}: " + " | ".join(out), flush=True)
