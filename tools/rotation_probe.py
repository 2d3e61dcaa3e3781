"""Why do six different 4096^3 GEMMs in rotation take ~58 us each when one of them repeated takes ~52?  Same kernel (int8 A@B.T), operands and
outputs rotated over N buffer sets: N * 64 MiB against the 256 MB Infinity Cache.  Calls the C ABI directly so that the output buffer is ours."""
import sys
sys.path[:0] = ["/root/repo/quantized-training_amd", "/root/repo"]
import torch

from gn_kernels import _cabi

L = _cabi.load()
S = 4096
dev = "cuda"


def mk(n):
    return [(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=dev), torch.randint(-128, 127, (S, S), dtype=torch.int8, device=dev),
             torch.empty(S, S, dtype=torch.bfloat16, device=dev)) for _ in range(n)]


rs = torch.randn(S, 1, device=dev).div(128).bfloat16()
cs = torch.randn(1, S, device=dev).div(128).bfloat16()
st = torch.cuda.current_stream().cuda_stream


def call(A, B, C):
    rc = L.gnk_scaled_mm(A.data_ptr(), 3, B.data_ptr(), 3, C.data_ptr(), 0, rs.data_ptr(), cs.data_ptr(), 0, 0, 0, S, S, S, S, 1, 1, S, st)
    assert rc == 0, L.gnk_last_error()


def region(fn, n=240):
    for _ in range(24):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


print("dtype codes:", _cabi.DTYPE_CODE)
sets = mk(8)
for n in (1, 2, 3, 4, 6, 8):
    i = [0]

    def all_rot():
        A, B, C = sets[i[0] % n]; i[0] += 1
        call(A, B, C)

    def c_rot():
        A, B, _ = sets[0]; C = sets[i[0] % n][2]; i[0] += 1
        call(A, B, C)

    def ab_rot():
        A, B, _ = sets[i[0] % n]; C = sets[0][2]; i[0] += 1
        call(A, B, C)

    print(f"{n} buffer sets ({n * 64} MiB): everything rotated {region(all_rot):5.1f} us | only C rotated {region(c_rot):5.1f} | only A, B rotated {region(ab_rot):5.1f}", flush=True)

# ---- is it the CODE that rotates?  Same buffers, kernels alternated
A, B, C = sets[0]
Bn = B.T.contiguous()                      # (K, N) row-major = N-major B for the A@B kernel
A8 = torch.randn(S, S, device=dev).to(torch.float8_e4m3fn)
B8 = torch.randn(S, S, device=dev).to(torch.float8_e4m3fn)
C2 = sets[1][2]


def kk():
    call(A, B, C)


def km():
    rc = L.gnk_scaled_mm(A.data_ptr(), 3, Bn.data_ptr(), 3, C2.data_ptr(), 0, rs.data_ptr(), cs.data_ptr(), 0, 0, 0, S, S, S, S, 1, S, 1, st)
    assert rc == 0


def kk8():
    rc = L.gnk_scaled_mm(A8.data_ptr(), 4, B8.data_ptr(), 4, C2.data_ptr(), 0, rs.data_ptr(), cs.data_ptr(), 0, 0, 0, S, S, S, S, 1, 1, S, st)
    assert rc == 0


small = torch.zeros(64, 64, device=dev)


def tiny():
    small.add_(1.0)


def seq(*fns):
    def f():
        for g in fns:
            g()
    return f


for label, fns in (("int8 KK x2", (kk, kk)), ("int8 KM x2", (km, km)), ("fp8 KK x2", (kk8, kk8)), ("int8 KK, int8 KM", (kk, km)), ("int8 KK, fp8 KK", (kk, kk8)),
                   ("int8 KK, tiny elementwise kernel", (kk, tiny)), ("int8 KK, tiny, int8 KK, tiny", (kk, tiny, kk, tiny))):
    t = region(seq(*fns), n=120)
    print(f"{label}: {t:6.1f} us per sequence", flush=True)

# ---- the bench step at C-ABI level (six kernels, six operand sets, fixed outputs) against the same step through the torch ops
import bench
import gn_kernels
cases = bench.make_inputs(S, dev)
keys = [(d, l) for d in ("int8", "fp8") for l in bench.LAYOUTS]
outs = [torch.empty(S, S, dtype=torch.bfloat16, device=dev) for _ in keys]
one_out = outs[0]


def abi(k, Cbuf):
    a, b, r, c = cases[k]
    code = 3 if a.dtype is torch.int8 else 4
    rc = L.gnk_scaled_mm(a.data_ptr(), code, b.data_ptr(), code, Cbuf.data_ptr(), 0, r.data_ptr(), c.data_ptr(), 0, 0, 0, S, S, S,
                         a.stride(0), a.stride(1), b.stride(0), b.stride(1), st)
    assert rc == 0


def step_abi_fixed():
    for k, o in zip(keys, outs):
        abi(k, o)


def step_abi_one_out():
    for k in keys:
        abi(k, one_out)


def step_ops():
    for k in keys:
        gn_kernels.scaled_mm(*cases[k])


def step_ops_keep():
    keep = [gn_kernels.scaled_mm(*cases[k]) for k in keys]
    return keep


for label, fn in (("C ABI, six outputs", step_abi_fixed), ("C ABI, one output buffer for all six", step_abi_one_out), ("torch ops, result dropped at once", step_ops),
                  ("torch ops, six results alive until the step ends", step_ops_keep)):
    t = region(fn, n=60)
    print(f"six-GEMM step, {label}: {t / 6:6.1f} us per GEMM", flush=True)
