"""SDPA fwd+bwd at the Llama-1B attention shape with the model's layout (projections give (B, S, H*D); q/k/v are transposed views),
flash vs memory-efficient backend, eager and compiled."""
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
xq, xk, xv = (torch.randn(B, S, H * D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
go = torch.randn(B, S, H * D, device="cuda", dtype=torch.bfloat16)


def attn(xq, xk, xv):
    q, k, v = (t.view(B, S, H, D).transpose(1, 2) for t in (xq, xk, xv))
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    return o.transpose(1, 2).reshape(B, S, H * D)


for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION)):
    with sdpa_kernel(be):
        ms = timeit(lambda: attn(xq, xk, xv).backward(go))
        print(f"eager    {name:14s}: fwd+bwd {ms:6.2f} ms", flush=True)
for name, fl in (("flash", True), ("mem_efficient", False)):
    torch._dynamo.reset()
    torch.backends.cuda.enable_flash_sdp(fl)
    torch.backends.cuda.enable_mem_efficient_sdp(not fl)
    c = torch.compile(attn)
    ms = timeit(lambda: c(xq, xk, xv).backward(go))
    print(f"compiled {name:14s}: fwd+bwd {ms:6.2f} ms", flush=True)


def attn_ctx(xq, xk, xv):
    q, k, v = (t.view(B, S, H, D).transpose(1, 2) for t in (xq, xk, xv))
    with sdpa_kernel(SDPBackend.EFFICIENT_ATTENTION):
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    return o.transpose(1, 2).reshape(B, S, H * D)


def attn_op(xq, xk, xv):
    q, k, v = (t.view(B, S, H, D).transpose(1, 2) for t in (xq, xk, xv))
    o = torch.ops.aten._scaled_dot_product_efficient_attention(q, k, v, None, True, 0.0, True)[0]
    return o.transpose(1, 2).reshape(B, S, H * D)


torch.backends.cuda.enable_flash_sdp(True)
torch.backends.cuda.enable_mem_efficient_sdp(True)
for name, fn in (("ctx manager", attn_ctx), ("aten efficient op", attn_op)):
    try:
        torch._dynamo.reset()
        c = torch.compile(fn)
        ms = timeit(lambda: c(xq, xk, xv).backward(go))
        print(f"compiled {name:18s}: fwd+bwd {ms:6.2f} ms", flush=True)
        ref = attn(xq, xk, xv)
        print("   max |diff| vs default:", (c(xq, xk, xv).float() - ref.float()).abs().max().item())
    except Exception as exc:
        print(f"compiled {name}: failed {type(exc).__name__}: {str(exc)[:300]}")
