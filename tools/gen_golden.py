#!/usr/bin/env python3
"""Generate golden vectors for the hot path by IMPORTING THE REFERENCE in the build container.

Run from the repo root (CPU only; the reference's Triton kernels run under TRITON_INTERPRET=1,
bypassing the autotuner with ``kernel.fn[grid]``):

    python tools/gen_golden.py

Writes small ``tests/golden/*.npz`` fixtures (inputs + expected outputs; bf16/fp16 stored as their
int16 bit patterns, fp8 as uint8).  Only these data files travel to the GPU box -- the reference
does not.  The interpreter truncates fp32 -> bf16 on store (an interpreter artefact), so every
kernel golden is taken with an fp32 output tensor; bf16 expectations are RNE casts of that.
"""
import os
import sys
from pathlib import Path

os.environ["TRITON_INTERPRET"] = "1"
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = Path("/root/reference")
sys.path[:0] = [str(REF / "kernels"), str(REF)]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import triton.language as tl  # noqa: E402

import gn_kernels  # noqa: E402  (the reference's package)
from gn_kernels import triton_mm as ref_mm  # noqa: E402
from subclasses.int8 import quantize_int8 as ref_quantize_int8  # noqa: E402
from subclasses import mixed_precision as ref_mp  # noqa: E402

assert Path(gn_kernels.__file__).is_relative_to(REF), gn_kernels.__file__

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def bits(t: torch.Tensor) -> np.ndarray:
    """numpy view of a tensor's storage order-preserving logical values (contiguous copy)."""
    t = t.detach().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy()
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(torch.uint8).numpy()
    return t.numpy()


def to_layout(x: torch.Tensor, column_major: bool) -> torch.Tensor:
    return x.T.contiguous().T if column_major else x.contiguous()


# ------------------------------------------------------------------------------- G1 quantize
def gen_quantize():
    g = torch.Generator().manual_seed(1234)
    out = {}
    x = torch.randn(64, 256, generator=g)
    x[3] = 0.0                                   # all-zero row -> scale 0, data 0
    x[5] = 0.0
    x[5, 17] = -3.25                             # single non-zero (negative max)
    x[7] = torch.arange(256).float().sub(127.5)  # exact .5 ties after scaling by 1.0 ...
    x[7, 0] = -127.0
    x[7, 255] = 127.0                            # ... amax 127 -> scale exactly 1
    x[9, :] = 1e-30                              # below eps clip
    x[11] = x[11].abs().neg()                    # all negative
    x[13, ::2] = 2.0 ** -3                       # many equal maxima
    for name, dt in [("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)]:
        xi = x.to(dt)
        q, s = ref_quantize_int8(xi, dim=1)
        out[f"x_{name}"], out[f"q_{name}"], out[f"s_{name}"] = bits(xi), q.numpy(), bits(s)
        # dim=0 on the same storage (column-wise absmax; result keeps x's strides)
        q0, s0 = ref_quantize_int8(xi, dim=0)
        assert q0.stride() == xi.stride()
        out[f"q_dim0_{name}"], out[f"s_dim0_{name}"] = q0.numpy(), bits(s0)
        # transposed VIEW with dim=1 (what _dynamic_int8_mm does with B.T): output keeps the view's strides
        qt, st = ref_quantize_int8(xi.T, dim=1)
        assert qt.stride() == xi.T.stride() and tuple(st.shape) == (256, 1)
        assert torch.equal(qt.T, q0) and torch.equal(st.view(-1), s0.view(-1))
    # stochastic rounding, with the reference's own torch.rand_like draw recorded for replay
    xb = x.bfloat16()
    torch.manual_seed(77)
    q_sr, s_sr = ref_quantize_int8(xb, True, dim=1)
    torch.manual_seed(77)
    noise = torch.rand_like(xb.float())
    chk = (xb.float() / (xb.float().abs().amax(1, keepdim=True) / 127).clip(1e-12) + noise).floor().clip(-128, 127)
    assert torch.equal(chk.to(torch.int8), q_sr), "noise replay does not reproduce the reference SR draw"
    out["sr_noise"], out["q_sr_bf16"], out["s_sr_bf16"] = noise.numpy(), q_sr.numpy(), bits(s_sr)
    # eps variant used by BitNet callers (bitnet.py:129)
    q_e, s_e = ref_quantize_int8(xb, dim=1, eps=1e-5)
    out["q_eps1e5_bf16"], out["s_eps1e5_bf16"] = q_e.numpy(), bits(s_e)
    np.savez_compressed(OUT / "g1_quantize_int8.npz", **out)
    print("G1 ok", {k: v.shape for k, v in out.items() if k.startswith("q_")})


# ------------------------------------------------------------------------------- kernel runners
def run_matmul_i32(A, B):
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, dtype=torch.int32)
    BM, BN, BK = 32, 32, 64
    grid = (-(-M // BM) * -(-N // BN),)
    ref_mm._matmul_kernel.fn[grid](A, B, C, M, N, K, *A.stride(), *B.stride(), *C.stride(),
                                   ACC_DTYPE=tl.int32, EVEN_K=True, BLOCK_M=BM, BLOCK_N=BN, BLOCK_K=BK)
    return C


def run_scaled(A, B, rs, cs):
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, dtype=torch.float32)
    BM, BN, BK = 32, 32, 64
    grid = (-(-M // BM) * -(-N // BN),)
    ref_mm._scaled_mm_kernel.fn[grid](A, B, C, rs, cs, M, N, K, *A.stride(), *B.stride(), *C.stride(),
                                      BLOCK_M=BM, BLOCK_N=BN, BLOCK_K=BK,
                                      ACC_DTYPE=tl.int32 if A.dtype == torch.int8 else tl.float32,
                                      EVEN_K=True, COL_SCALE_SCALAR=cs.numel() == 1)
    return C


def run_tile_scaled(A, B, sA, sB):
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, dtype=torch.float32)
    QM, QN, QK = M // sA.shape[0], N // sB.shape[1], K // sA.shape[1]
    BM, BN = 32, 32
    grid = (-(-M // BM) * -(-N // BN),)
    ref_mm._tile_scaled_mm_kernel.fn[grid](A, B, C, sA, sB, M, N, K, *A.stride(), *B.stride(), *C.stride(),
                                           *sA.stride(), *sB.stride(),
                                           QUANT_BLOCK_M=QM, QUANT_BLOCK_N=QN, QUANT_BLOCK_K=QK,
                                           BLOCK_M=BM, BLOCK_N=BN, BLOCK_K=QK,
                                           ACC_DTYPE=tl.int32 if A.dtype == torch.int8 else tl.float32, EVEN_K=True)
    return C


LAYOUTS = {"nt": (False, True), "nn": (False, False), "tn": (True, False), "tt": (True, True)}


# ------------------------------------------------------------------------------- G2/G3 int8
def gen_int8_mm():
    g = torch.Generator().manual_seed(2025 * 2024)
    M, N, K = 96, 160, 256
    A = torch.randint(-128, 127, (M, K), dtype=torch.int8, generator=g)
    B = torch.randint(-128, 127, (K, N), dtype=torch.int8, generator=g)
    A[0, :] = -128
    B[:, 0] = -128                                 # extreme accumulator: 128*128*K
    out = {"A": A.numpy(), "B": B.numpy()}
    ref = torch._int_mm(A, B)
    assert torch.equal(ref, A.int() @ B.int())
    for name, (acm, bcm) in LAYOUTS.items():
        C = run_matmul_i32(to_layout(A, acm), to_layout(B, bcm))
        assert torch.equal(C, ref), name         # the kernel is layout-agnostic and == torch._int_mm
    out["C_i32"] = ref.numpy()

    rs = {dt: torch.randn(M, 1, generator=g).div(128).to(dt) for dt in (torch.bfloat16, torch.float16, torch.float32)}
    cs = {dt: torch.randn(1, N, generator=g).div(128).to(dt) for dt in (torch.bfloat16, torch.float16, torch.float32)}
    for name, dt in [("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)]:
        out[f"rs_{name}"], out[f"cs_{name}"] = bits(rs[dt]), bits(cs[dt])
        for lname, (acm, bcm) in LAYOUTS.items():
            C = run_scaled(to_layout(A, acm), to_layout(B, bcm), rs[dt], cs[dt])
            if lname == "nt":
                out[f"C_scaled_{name}"] = C.numpy()
                # pin the multiply order: (f32(acc) * row) * col
                exp = (ref.float() * rs[dt].float()) * cs[dt].float()
                assert torch.equal(C, exp)
            else:
                assert np.array_equal(C.numpy(), out[f"C_scaled_{name}"]), lname
        Cs = run_scaled(A, to_layout(B, True), rs[dt], cs[dt][:, :1].contiguous())
        out[f"C_scaled_scalar_{name}"] = Cs.numpy()
    np.savez_compressed(OUT / "g2_g3_int8_mm.npz", **out)
    print("G2/G3 ok")


# ------------------------------------------------------------------------------- G4 tile-scaled int8
def gen_tile_scaled_int8():
    g = torch.Generator().manual_seed(4)
    out = {}
    # DeepSeek style (1,128) x (128,128) as in benchmark_mm.py:87-89
    M, N, K = 64, 256, 512
    A = torch.randint(-128, 127, (M, K), dtype=torch.int8, generator=g)
    B = torch.randint(-128, 127, (K, N), dtype=torch.int8, generator=g)
    sA = torch.randn(M, K // 128, generator=g).div(128).bfloat16()
    sB = torch.randn(K // 128, N // 128, generator=g).div(128).bfloat16()
    ref = None
    for lname, (acm, bcm) in LAYOUTS.items():
        C = run_tile_scaled(to_layout(A, acm), to_layout(B, bcm), to_layout(sA, acm), to_layout(sB, bcm))
        if ref is None:
            ref = C
        assert torch.equal(C, ref), lname
    out.update(ds_A=A.numpy(), ds_B=B.numpy(), ds_sA=bits(sA), ds_sB=bits(sB), ds_C=ref.numpy())
    # Jetfire style 32x32 blocks on both operands, fp32 scales
    M, N, K = 64, 128, 256
    A = torch.randint(-128, 127, (M, K), dtype=torch.int8, generator=g)
    B = torch.randint(-128, 127, (K, N), dtype=torch.int8, generator=g)
    sA = torch.randn(M // 32, K // 32, generator=g).div(128)
    sB = torch.randn(K // 32, N // 32, generator=g).div(128)
    C = run_tile_scaled(A, to_layout(B, True), sA, sB)
    out.update(jf_A=A.numpy(), jf_B=B.numpy(), jf_sA=sA.numpy(), jf_sB=sB.numpy(), jf_C=C.numpy())
    np.savez_compressed(OUT / "g4_tile_scaled_int8.npz", **out)
    print("G4 ok")


# ------------------------------------------------------------------------------- G5 fp8
def gen_fp8():
    g = torch.Generator().manual_seed(5)
    M, N, K = 64, 128, 256
    out = {}
    Af = torch.randn(M, K, generator=g)
    Bf = torch.randn(K, N, generator=g)
    rs = torch.randn(M, 1, generator=g).div(128).bfloat16()
    cs = torch.randn(1, N, generator=g).div(128).bfloat16()
    sA = torch.randn(M, K // 128, generator=g).div(128).bfloat16()
    sB = torch.randn(K // 128, N // 128, generator=g).div(128).bfloat16()
    out.update(rs=bits(rs), cs=bits(cs), sA=bits(sA), sB=bits(sB))
    f8 = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}
    for an, bn in [("e4m3", "e4m3"), ("e5m2", "e5m2"), ("e4m3", "e5m2")]:
        A = Af.to(f8[an])
        B = to_layout(Bf.to(f8[bn]), True)
        tag = f"{an}_{bn}"
        out[f"A_{tag}"], out[f"B_{tag}"] = bits(A), bits(B)
        out[f"C_scaled_{tag}"] = run_scaled(A, B, rs, cs).numpy()
        out[f"C_tile_{tag}"] = run_tile_scaled(A, B, sA, sB).numpy()
        out[f"C_plain_{tag}"] = (A.float() @ B.float()).numpy()   # benchmark_mm.py:128
    np.savez_compressed(OUT / "g5_fp8.npz", **out)
    print("G5 ok")


# ------------------------------------------------------------------------------- G6 subclass fwd/bwd
def gen_mixed_precision_linear():
    calls = []

    @torch.library.impl(gn_kernels._lib.lib, "scaled_mm", "CPU")
    def _(A, B, row_scale, col_scale):
        calls.append([list(A.shape), list(A.stride()), list(B.shape), list(B.stride()),
                      list(row_scale.shape), list(col_scale.shape)])
        return run_scaled(A, B, row_scale, col_scale).to(row_scale.dtype)   # RNE cast of the fp32 kernel result

    out = {}
    for tag, cfg_kwargs, bias in [("all", {}, True), ("fwd_only", dict(grad_input=False, grad_weight=False), False)]:
        torch.manual_seed(0)
        lin = torch.nn.Linear(256, 512, bias=bias).bfloat16()
        x = torch.randn(4, 64, 256).bfloat16().requires_grad_()
        gy = torch.randn(4, 64, 512).bfloat16()
        out[f"{tag}_W"], out[f"{tag}_x"], out[f"{tag}_gy"] = bits(lin.weight), bits(x), bits(gy)
        if bias:
            out[f"{tag}_b"] = bits(lin.bias)
        ref_mp.convert_mixed_precision(lin, config=ref_mp.MixedPrecisionConfig(**cfg_kwargs))
        calls.clear()
        y = lin(x)
        y.backward(gy)
        out[f"{tag}_y"], out[f"{tag}_dx"] = bits(y), bits(x.grad)
        gw = lin.weight.grad
        gw = gw._data if hasattr(gw, "_data") else gw
        out[f"{tag}_dW"] = bits(gw)
        if bias:
            out[f"{tag}_db"] = bits(lin.bias.grad)
        out[f"{tag}_calls"] = np.array(calls, dtype=np.int64).reshape(len(calls), -1)
        print("G6", tag, "scaled_mm calls:", calls)
    np.savez_compressed(OUT / "g6_mixed_precision_linear.npz", **out)
    print("G6 ok")


if __name__ == "__main__":
    gen_quantize()
    gen_int8_mm()
    gen_tile_scaled_int8()
    gen_fp8()
    gen_mixed_precision_linear()
    total = sum(p.stat().st_size for p in OUT.glob("*.npz"))
    print(f"wrote {OUT} ({total / 1024:.0f} KiB)")
