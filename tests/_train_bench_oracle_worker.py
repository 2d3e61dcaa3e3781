"""TEST-ONLY launcher of train_bench.py on CPU tensors: registers the CPU ORACLE as the gn_kernels ops' CPU backend and routes host
tensors through the subclasses' device branch, so the INT8 mixed-precision harness (DDP over gloo, torch.compile) can be exercised
without a GPU.  The product never does this: its ops have Meta and device implementations only."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]

import torch  # noqa: E402

import gn_kernels  # noqa: E402,F401
import oracle  # noqa: E402
from gn_kernels import quantize as qmod  # noqa: E402

impl = torch.library.Library("gn_kernels", "IMPL")


def _sr(sr, rng):
    return dict(seed=int(rng.item()), offset=0) if sr else {}


impl.impl("scaled_mm", lambda A, B, sa, sb: oracle.scaled_mm(A, B, sa, sb), "CPU")
impl.impl("scaled_mm_t", lambda A, B, sa, sb: oracle.scaled_mm(A, B, sa, sb).T.contiguous(), "CPU")
impl.impl("quantize_int8", lambda x, dim, sr, eps, rng=None: oracle.quantize_int8(x, sr, dim=dim, eps=eps, **_sr(sr, rng)), "CPU")
impl.impl("quantize_int8_rowwise_colmax",
          lambda x, sr, eps, rng=None: (*oracle.quantize_int8(x, sr, dim=1, eps=eps, **_sr(sr, rng)), x.float().abs().amax(0)), "CPU")
impl.impl("quantize_int8_colwise_given", lambda x, am, sr, eps, rng=None: oracle.quantize_int8(x, sr, dim=0, eps=eps, **_sr(sr, rng)), "CPU")
qmod.on_device = lambda t: True

import train_bench  # noqa: E402

if __name__ == "__main__":
    out = train_bench.run(train_bench.parse_args(sys.argv[1:]))
    if out is not None:
        print(json.dumps(out), flush=True)
