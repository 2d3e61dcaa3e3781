import sys
sys.path[:0] = ["/root/repo/quantized-training_amd", "/root/repo"]
import torch, bench, gn_kernels
S = 4096
lay = lambda x, cm: x.T.contiguous().T if cm else x.contiguous()
for kind in ("i8", "f8"):
    for (QM, QN) in ((1, 1), (128, 2), (1, 128)):
        out = []
        for name, (acm, bcm) in {"A@B.T": (False, True), "A@B": (False, False), "A.T@B": (True, False)}.items():
            if kind == "i8":
                A = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), acm); B = lay(torch.randint(-128, 127, (S, S), dtype=torch.int8, device="cuda"), bcm)
            else:
                A = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), acm); B = lay(torch.randn(S, S, device="cuda").to(torch.float8_e4m3fn), bcm)
            sA = torch.randn(S // QM, S // 128, device="cuda").div(128).bfloat16()
            sB = torch.randn(S // 128, S // QN, device="cuda").div(128).bfloat16()
            avg, _ = bench.time_kernel(lambda: gn_kernels.scaled_mm(A, B, sA, sB), iters=30)
            out.append(f"{name} {avg * 1e3:6.1f} us {2 * S ** 3 / avg * 1e-9:5.0f} T")
        print(f"{kind} blocks ({QM}, {QN}, 128): " + " | ".join(out), flush=True)
