"""Average launch duration of one GEMM three ways (GPU box): HIP events around EVERY launch (bench.py's method in rounds 1-2: each pair of
event records adds its own packets; tools/probe_launch shows 6.0 us for an empty kernel against 2.3-2.6 us back to back), ONE event pair
around N back-to-back launches divided by N, and a HIP-graph replay of N launches."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import bench
import gn_kernels

S = 4096
cases = bench.make_inputs(S, "cuda")
for key in (("int8", "A@B.T"), ("int8", "A@B"), ("int8", "A.T@B"), ("fp8", "A@B.T")):
    a = cases[key]
    fn = lambda: gn_kernels.scaled_mm(*a)
    avg, med = bench.time_kernel(fn, iters=50)
    for _ in range(10):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 200
    torch.cuda.synchronize()
    s.record()
    for _ in range(N):
        fn()
    e.record()
    torch.cuda.synchronize()
    region = s.elapsed_time(e) / N
    graph = bench.time_kernel_graph(fn, per_graph=50, replays=5)
    ops = 2 * S ** 3
    print(f"{key}: per-launch events avg {avg * 1e3:6.2f} us ({ops / avg * 1e-9:6.0f} T)  median {med * 1e3:6.2f} | one event pair around {N} launches {region * 1e3:6.2f} us ({ops / region * 1e-9:6.0f} T)"
          f" | graph replay {graph * 1e3:6.2f} us ({ops / graph * 1e-9:6.0f} T)", flush=True)
