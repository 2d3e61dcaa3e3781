"""Host-side cost per op call (no synchronisation inside the timed loop; small problem so the GPU keeps up)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi, mm
from subclasses import quantize_int8

L = _cabi.load()
A = torch.randint(-128, 127, (256, 256), dtype=torch.int8, device="cuda")
Bt = torch.randint(-128, 127, (256, 256), dtype=torch.int8, device="cuda")
rs = torch.randn(256, 1, device="cuda").bfloat16()
cs = torch.randn(1, 256, device="cuda").bfloat16()
x = torch.randn(256, 256, device="cuda").bfloat16()
C = torch.empty(256, 256, device="cuda", dtype=torch.bfloat16)


def t(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


st = _cabi.stream_of(A)
code = _cabi.DTYPE_CODE
print(f"gn_kernels.scaled_mm (wrapper + dispatcher + impl): {t(lambda: gn_kernels.scaled_mm(A, Bt.T, rs, cs)):6.1f} us")
print(f"torch.ops.gn_kernels.scaled_mm (dispatcher + impl): {t(lambda: torch.ops.gn_kernels.scaled_mm(A, Bt.T, rs, cs)):6.1f} us")
print(f"impl function called directly:                      {t(lambda: mm._scaled_mm_hip(A, Bt.T, rs, cs)):6.1f} us")
print(f"raw ctypes call only (preallocated C, cached stream): {t(lambda: L.gnk_scaled_mm(A.data_ptr(), 3, Bt.data_ptr(), 3, C.data_ptr(), 0, rs.data_ptr(), cs.data_ptr(), 0, 0, 0, 256, 256, 256, 256, 1, 1, 256, st)):6.1f} us")
print(f"torch.empty((256,256)):                             {t(lambda: torch.empty((256, 256), device='cuda', dtype=torch.bfloat16)):6.1f} us")
print(f"_cabi.stream_of:                                    {t(lambda: _cabi.stream_of(A)):6.1f} us")
print(f"raw stream query torch._C._cuda_getCurrentRawStream: {t(lambda: torch._C._cuda_getCurrentRawStream(0)):6.1f} us")
print(f"Bt.T view:                                          {t(lambda: Bt.T):6.1f} us")
print(f"quantize_int8(x, dim=1):                            {t(lambda: quantize_int8(x, dim=1)):6.1f} us")
print(f"quantize_int8(x, dim=0):                            {t(lambda: quantize_int8(x, dim=0)):6.1f} us")
print(f"torch.mm bf16 256 (ATen reference point):           {t(lambda: torch.mm(x, x)):6.1f} us")
