"""HIP-graph-replay timing of the BitNet weight kernels and of one BitNet Linear forward, beside the ATen chains of the reference
(subclasses/bitnet.py:110-118, :185-195) on the same device tensors, a bf16 F.linear and the INT8 mixed-precision forward."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]
import torch
import torch.nn.functional as F

from gn_kernels import bitnet as bop
from subclasses import bitnet as bn
from subclasses import convert_mixed_precision
from subclasses.mixed_precision import set_quantize_plan
from tools.fused_quant_probe import graph_time


def aten_scale(w):
    return w.float().abs().mean()


def aten_quant(w, s):
    return (w.float() / s.clip(1e-5)).round().clip(-1, 1).to(torch.int8)


def aten_pack(x):
    return (x[:, ::4] << 6) | ((x[:, 1::4] & 3) << 4) | ((x[:, 2::4] & 3) << 2) | (x[:, 3::4] & 3)


for shape in [(2048, 2048), (5632, 2048), (4096, 4096), (11008, 4096), (8192, 8192)]:
    w = (torch.randn(shape, device="cuda") * 0.02).bfloat16()
    n = w.numel()
    s = bop.bitnet_absmean(w)
    q = bop.bitnet_quantize_weight(w, s)
    p = bop.bitnet_quantize_weight(w, s, packed=True)
    t_s = graph_time(lambda: bop.bitnet_absmean(w))
    t_q = graph_time(lambda: bop.bitnet_quantize_weight(w, s))
    t_p = graph_time(lambda: bop.bitnet_quantize_weight(w, s, packed=True))
    t_u = graph_time(lambda: bop.bitnet_unpack_i2(p))
    a_s = graph_time(lambda: aten_scale(w))
    a_q = graph_time(lambda: aten_quant(w, s))
    a_p = graph_time(lambda: aten_pack(q))
    print(f"{shape[0]}x{shape[1]}: absmean {t_s:6.1f} us ({2 * n / t_s / 1e6:5.2f} TB/s; ATen {a_s:6.1f}) | quantize {t_q:6.1f} us ({3 * n / t_q / 1e6:5.2f} TB/s; ATen {a_q:6.1f})"
          f" | quantize+pack {t_p:6.1f} us ({2.25 * n / t_p / 1e6:5.2f} TB/s; ATen quantize + pack {a_q + a_p:6.1f}) | unpack {t_u:6.1f} us ({1.25 * n / t_u / 1e6:5.2f} TB/s)", flush=True)

set_quantize_plan(cache=False)      # the replay quantizes the same x every time: the cache would hide that pass
print("one Linear forward, T tokens x (in -> out), bf16 activations:")
for T, fin, fout in [(8192, 1024, 4096), (8192, 4096, 1024), (32768, 2048, 5632), (32768, 5632, 2048)]:
    lin = torch.nn.Linear(fin, fout, bias=False, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(T, fin, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        t_bf = graph_time(lambda: lin(x))
        import copy
        l_bn = bn.convert_bitnet(copy.deepcopy(lin))
        l_mp = convert_mixed_precision(copy.deepcopy(lin))
        t_bn = graph_time(lambda: l_bn(x))
        t_mp = graph_time(lambda: l_mp(x))
    flop = 2 * T * fin * fout
    print(f"  {T} x ({fin} -> {fout}): bf16 {t_bf:7.1f} us ({flop / t_bf / 1e6:6.0f} TFLOP/s) | INT8 mixed-precision {t_mp:7.1f} us | BitNet {t_bn:7.1f} us ({flop / t_bn / 1e6:6.0f} TOP/s)", flush=True)
