"""does two-level accumulation bring the UNSCALED fp8 product (fp8_mm / _triton_mm, benchmark_mm.py:128-129) inside the
reference's unwidened assert_close (bf16 defaults, rtol 1.6e-2, atol 1e-5) at 4096^3?  The tile-scaled kernel with unit scales IS a two-level
accumulation: a fresh MFMA chain per QK-block, added into an fp32 register set with v_add_f32.  Run on the GPU box; prints one JSON line per case."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels

torch.manual_seed(2025 * 2024)
S = 4096
A = torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn)
B = torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn).T.contiguous().T
want64 = A.double() @ B.double()
want = (A.float() @ B.float())
wb = want.bfloat16().float()


def report(name, out):
    o = out.float()
    fail = int(((o - wb).abs() > 1.6e-2 * wb.abs() + 1e-5).sum().item())
    err = (out.double() - want64).abs()
    print(json.dumps({"case": name, "unwidened_failing": fail, "of": S * S, "max_abs_err_vs_fp64": float(err.max()), "mean_abs_err_vs_fp64": float(err.mean())}), flush=True)


report("fp32 torch.mm rounded to bf16 (the reference's own comparison value)", want.bfloat16())
report("plain fp8 MFMA chain over K=4096 (bf16 out)", gn_kernels._triton_mm(A, B, out_dtype=torch.bfloat16))
report("plain fp8 MFMA chain over K=4096 (fp32 out)", gn_kernels._triton_mm(A, B, out_dtype=torch.float32))
for qk in (32, 64, 128, 256, 512, 1024):
    sa = torch.ones(S, S // qk, device="cuda")
    sb = torch.ones(S // qk, S // 128, device="cuda")
    report(f"two-level: fresh chain per {qk} k + fp32 add (fp32 out)", gn_kernels.scaled_mm(A, B, sa, sb))
    report(f"two-level: fresh chain per {qk} k + fp32 add (bf16 out)", gn_kernels.scaled_mm(A, B, sa.bfloat16(), sb.bfloat16()))
