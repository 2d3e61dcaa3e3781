"""A/B timing of the GEMM tile variants at M=N=4096 (GPU box): us and TOPS per variant, layout and K."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch

import gn_kernels
from gn_kernels import _cabi


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(s, e))
    return ts[len(ts) // 2] * 1e3


def lay(x, cm):
    return x.T.contiguous().T if cm else x.contiguous()


def main(ks=(256, 1024, 4096, 8192), variants=None):
    L = _cabi.load()
    S = 4096
    rs = torch.randn(S, 1, device="cuda").div(128).bfloat16()
    cs = torch.randn(1, S, device="cuda").div(128).bfloat16()
    variants = variants or {"pp2": 0, "pp8": 4, "w4": 5}
    import os
    if os.environ.get("GNK_CMP_VARIANTS"):   # e.g. GNK_CMP_VARIANTS=w4,pp8
        variants = {k: v for k, v in variants.items() if k in os.environ["GNK_CMP_VARIANTS"].split(",")}
    for K in ks:
        for name, (acm, bcm) in {"nt": (False, True), "nn": (False, False), "tn": (True, False)}.items():
            A = lay(torch.randint(-128, 127, (S, K), dtype=torch.int8, device="cuda"), acm)
            B = lay(torch.randint(-128, 127, (K, S), dtype=torch.int8, device="cuda"), bcm)
            A8 = lay(torch.randn(S, K, device="cuda").to(torch.float8_e4m3fn), acm)
            B8 = lay(torch.randn(K, S, device="cuda").to(torch.float8_e4m3fn), bcm)
            out = []
            for vn, v in variants.items():
                L.gnk_set_gemm_variant(v)
                t = timeit(lambda: gn_kernels.scaled_mm(A, B, rs, cs))
                out.append(f"i8 {vn} {t:7.1f}us {2*S*S*K/t/1e6:6.0f}T")
                if True:
                    t = timeit(lambda: gn_kernels.scaled_mm(A8, B8, rs, cs))
                    out.append(f"f8 {vn} {t:7.1f}us {2*S*S*K/t/1e6:6.0f}T")
            L.gnk_set_gemm_variant(-1)
            print(f"K={K:5d} {name}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main(ks=tuple(int(k) for k in sys.argv[1:]) or (256, 1024, 4096, 8192))
