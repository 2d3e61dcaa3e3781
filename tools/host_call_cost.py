"""Host cost of one gn_kernels.scaled_mm call (dispatcher -> C++ shim -> C ABI -> hipLaunchKernel) and of the bench step's six calls:
wall time of N calls on 128^3 operands (GPU time negligible), and the GPU-side region average of the six 4096^3 GEMMs of bench.py's step."""
import sys
import time
sys.path[:0] = ["/root/repo/quantized-training_amd", "/root/repo"]
import torch

import bench
import gn_kernels

small = bench.make_inputs(128, "cuda")
a = small[("int8", "A@B.T")]
for _ in range(200):
    gn_kernels.scaled_mm(*a)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 2000
for _ in range(N):
    gn_kernels.scaled_mm(*a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host: {(t1 - t0) / N * 1e6:.1f} us per call issued, {(t2 - t0) / N * 1e6:.1f} us per call until the queue drained")

S = 4096
cases = bench.make_inputs(S, "cuda")
keys = [(d, l) for d in ("int8", "fp8") for l in bench.LAYOUTS]


def step():
    for k in keys:
        gn_kernels.scaled_mm(*cases[k])


for _ in range(10):
    step()
torch.cuda.synchronize()
for label, fn, per in (("six different GEMMs per step", step, 6),
                       ("same int8 A@B.T GEMM six times", lambda: [gn_kernels.scaled_mm(*cases[keys[0]]) for _ in range(6)], 6)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label}: GPU {e0.elapsed_time(e1) / 50 / per * 1e3:.1f} us per GEMM, host issue {(t1 - t0) / 50 / per * 1e6:.1f} us per GEMM")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(5):
            step()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10):
        g.replay()
    e1.record(s)
    torch.cuda.synchronize()
print(f"six different GEMMs per step, HIP-graph replay: {e0.elapsed_time(e1) / 50 / 6 * 1e3:.1f} us per GEMM")
