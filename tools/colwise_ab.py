"""A/B of the two pass-1 forms of the strided-reduction quantize (csrc/gnk_quantize.hip:launch_colwise), through the C ABI, HIP-graph
replay (20 calls per graph): per-block partial maxima (the workspace gnk_quantize_int8_workspace_bytes asks for) against the minimum
workspace (hipMemsetAsync + atomicMax)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "quantized-training_amd"))
from gn_kernels import _cabi  # noqa: E402

L = _cabi.load()
DEV = "cuda"


def bench(x, nbytes, full, reps=20, iters=20):
    P, Q = x.shape
    ws = torch.empty(full, dtype=torch.uint8, device=DEV)
    q = torch.empty_like(x, dtype=torch.int8)
    s = torch.empty(Q, dtype=x.dtype, device=DEV)
    code = _cabi.DTYPE_CODE[x.dtype]

    def call():
        rc = L.gnk_quantize_int8(x.data_ptr(), code, Q, P, 1, x.stride(0), q.data_ptr(), 1, q.stride(0), s.data_ptr(), 1e-12, 0, 0, 0,
                                 ws.data_ptr(), nbytes, _cabi.stream_of(x))
        assert rc == 0, L.gnk_last_error()

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                call()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(iters):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    # the same calls straight on a stream (no graph): the memset is then a fill kernel of its own, as in a training step
    e0.record()
    for _ in range(200):
        call()
    e1.record()
    torch.cuda.synchronize()
    return best, e0.elapsed_time(e1) * 1e3 / 200, q.clone(), s.clone()


for shape in [(1024, 1024), (4096, 1024), (1024, 4096), (8192, 1024), (8192, 4096), (2048, 2048), (5632, 2048), (2048, 5632), (4096, 4096), (16384, 8192), (32768, 2048)]:
    torch.manual_seed(0)
    x = torch.randn(shape, device=DEV, dtype=torch.bfloat16)
    P, Q = shape
    full = L.gnk_quantize_int8_workspace_bytes(Q, P, 1, x.stride(0))
    tp, sp_us, qp, sp = bench(x, full, full)
    ta, sa_us, qa, sa = bench(x, Q * 4, full)
    same = torch.equal(qp, qa) and torch.equal(sp.view(torch.int16), sa.view(torch.int16))
    print(f"{shape[0]:6d} x {shape[1]:5d} bf16: partial maxima {tp:7.2f} us   memset + atomicMax {ta:7.2f} us   ({ta / tp:.2f}x)   on a stream: {sp_us:7.2f} / {sa_us:7.2f} us ({sa_us / sp_us:.2f}x)  identical={same}", flush=True)
