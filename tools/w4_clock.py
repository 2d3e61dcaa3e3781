"""In-kernel clock of the one-wave-per-SIMD GEMM loop (GPU box; needs a stamp build: tools/w4_ablate.sh 16 [17 ...] and
GNK_LIBRARY=.../libgnk_w4abl16.so).  Per workgroup: cycles and 100 MHz ticks around the main loop -> GHz, cycles per 64-k stage."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import numpy as np
import torch

import gn_kernels
from gn_kernels import _cabi

L = _cabi.load()
L.gnk_debug_w4_stamps.argtypes = [ctypes.c_void_p]
S = 4096
for K in (4096, 8192):
    for name, (acm, bcm) in {"nt": (False, True), "nn": (False, False), "tn": (True, False)}.items():
        lay = lambda x, cm: x.T.contiguous().T if cm else x.contiguous()
        A = lay(torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda"), acm)
        B = lay(torch.randint(-128, 127, (K, S), dtype=torch.int8, device="cuda"), bcm)
        rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
        cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
        L.gnk_set_gemm_variant(5)
        for _ in range(400):     # ~25 ms of back-to-back launches so that the clock has settled
            gn_kernels.scaled_mm(A, B, rs, cs)
        torch.cuda.synchronize()
        L.gnk_set_gemm_variant(-1)
        buf = np.zeros(4096, dtype=np.uint64)
        assert L.gnk_debug_w4_stamps(buf.ctypes.data) == 0
        r = buf.reshape(512, 8)[:256].astype(np.int64)
        cyc, ticks = r[:, 1] - r[:, 0], r[:, 4] - r[:, 3]
        ghz = cyc / ticks * 0.1
        ns = K // 64
        us = lambda a, b: np.median(r[:, b] - r[:, a]) / 100.0
        first, last = r[:, 2].min(), r[:, 6].max()
        print(f"K={K} {name}: loop {np.median(ticks) / 100:6.1f} us  clock {np.median(ghz):.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f})  "
              f"{np.median(cyc) / ns:7.1f} cycles / stage ({np.median(ticks) * 10 / ns:6.1f} ns) | entry->loop {us(2, 3):5.2f}  loop->half0 stored {us(4, 5):5.2f}  "
              f"half1 {us(5, 6):5.2f}  | first entry -> last end {(last - first) / 100.0:6.1f} us, entry spread {(r[:, 2].max() - first) / 100.0:5.2f}", flush=True)
