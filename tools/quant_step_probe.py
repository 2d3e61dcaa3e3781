"""The two in-step quantize kernels at the Llama-1B activation shapes, cold (three input tensors in rotation: 400-1100 MB, beyond the
256 MB Infinity Cache), HIP-graph replayed: row-wise + column absmax (forward / dgrad side) and column-wise with the absmax given
(wgrad side).  TB/s = algorithmic 3 bytes per element.  A/B across library builds with tools/ab_libs.sh."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

from gn_kernels import quantize as qmod
from tools.fused_quant_probe import graph_time

for R, C in ((32768, 2048), (32768, 5632), (8192, 1024), (8192, 4096)):
    xs = [torch.randn(R, C, device="cuda").bfloat16() for _ in range(3)]
    cms = [qmod.quantize_int8_rowwise_colmax(x)[2] for x in xs]
    i = [0]

    def rc():
        i[0] = (i[0] + 1) % 3
        return qmod.quantize_int8_rowwise_colmax(xs[i[0]])

    def cg():
        i[0] = (i[0] + 1) % 3
        return qmod.quantize_int8_colwise_given(xs[i[0]], cms[i[0]])

    def rw():
        i[0] = (i[0] + 1) % 3
        return qmod.quantize_int8(xs[i[0]], dim=1)

    t_rc, t_cg, t_rw = graph_time(rc, iters=12), graph_time(cg, iters=12), graph_time(rw, iters=12)
    b = 3 * R * C
    print(f"{R}x{C}: row+colmax {t_rc:7.1f} us {b / t_rc * 1e-6:5.2f} TB/s | col given {t_cg:7.1f} us {b / t_cg * 1e-6:5.2f} TB/s | plain row {t_rw:7.1f} us {b / t_rw * 1e-6:5.2f} TB/s", flush=True)
