"""How long after an idle period does the six-GEMM step of bench.py reach its steady time?  One HIP event per step, from a cold start (inputs
just created, GPU idle) and again after 2 s of idleness."""
import sys
import time
sys.path[:0] = ["/root/repo/quantized-training_amd", "/root/repo"]
import torch

import bench
import gn_kernels

S = 4096
cases = bench.make_inputs(S, "cuda")
keys = [(d, l) for d in ("int8", "fp8") for l in bench.LAYOUTS]
gn_kernels.scaled_mm(*cases[keys[0]])
torch.cuda.synchronize()


def run(label, n=800):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        for k in keys:
            gn_kernels.scaled_mm(*cases[k])
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    edges = [0, 5, 10, 20, 40, 80, 160, 320, 480, 640, 800]
    print(label + ": " + "  ".join(f"[{a}-{b}) {sum(ts[a:b]) / (b - a):5.1f}" for a, b in zip(edges, edges[1:])) + "   us per step", flush=True)


time.sleep(2.0)
run("from idle        ")
time.sleep(2.0)
run("after 2 s of idle")
run("back to back     ")
