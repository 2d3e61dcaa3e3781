"""Event-timed (HIP-graph replay) cost of the quantization of one tensor along both dims: separate passes vs the fused pair."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

from gn_kernels import quantize as qmod


def graph_time(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    for shape in [(8192, 1024), (8192, 4096), (4096, 2048), (4096, 5632), (32768, 2048), (32768, 5632), (2048, 2048), (5632, 2048), (2048, 5632)]:
        x = torch.randn(shape, device="cuda").bfloat16()
        R, C = shape
        t_row = graph_time(lambda: qmod.quantize_int8(x, dim=1))
        t_col = graph_time(lambda: qmod.quantize_int8(x, dim=0))
        t_rc = graph_time(lambda: qmod.quantize_int8_rowwise_colmax(x))
        _, _, cm = qmod.quantize_int8_rowwise_colmax(x)
        t_cg = graph_time(lambda: qmod.quantize_int8_colwise_given(x, cm))
        mb = R * C / 1e6
        print(f"{R}x{C}: separate row {t_row:7.1f} + col {t_col:7.1f} = {t_row + t_col:7.1f} us | fused row+colmax {t_rc:7.1f} + col-given {t_cg:7.1f} = {t_rc + t_cg:7.1f} us"
              f" | {6 * mb / (t_rc + t_cg):5.2f} TB/s algorithmic (6 B/elem)", flush=True)


if __name__ == "__main__":
    main()
