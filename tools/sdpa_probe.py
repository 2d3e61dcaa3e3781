"""Which SDPA backend does the training harness get on this stack?  Llama-1B attention shape (16 x 32 heads x 2048 x 64, bf16, causal),
forward + backward, event-timed.  Attention is outside the scaled-matmul path; this only tells how much of the step it takes."""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B, H, S, D = 16, 32, 2048, 64
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
go = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16)
flops_f = 4 * B * H * S * S * D / 2


def run():
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    out.backward(go)


for lib in ("aotriton", "ck"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        print("preferred_rocm_fa_library ->", torch.backends.cuda.preferred_rocm_fa_library())
    except Exception as exc:
        print(f"{lib}: not available ({type(exc).__name__}: {str(exc)[:120]})")
        continue
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with sdpa_kernel(be):
                ms = timeit(run)
                with torch.no_grad():
                    msf = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))
            print(f"  {lib:9s} {name:14s}: fwd {msf:6.2f} ms ({flops_f / msf * 1e-9:6.0f} TFLOP/s)   fwd+bwd {ms:6.2f} ms ({3.5 * flops_f / ms * 1e-9:6.0f} TFLOP/s)", flush=True)
        except Exception as exc:
            print(f"  {lib:9s} {name:14s}: failed ({type(exc).__name__}: {str(exc)[:120]})")
