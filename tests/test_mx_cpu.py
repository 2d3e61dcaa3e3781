"""CPU checks of the MX path: the numpy oracle (oracle/mx_ref.py) and the host-side ATen formulas of gn_kernels.mx against
the goldens generated from the reference (tools/gen_golden_mx.py), bit for bit."""
import numpy as np
import pytest
import torch

from gn_kernels import mx
from oracle import mx_ref
from tests._util import load

FMTS = {"fp4": mx.FP4, "e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}


def _x(g, xname):
    if xname == "f32":
        return torch.from_numpy(g["x_f32"])
    return torch.from_numpy(g["x_bf16"]).view(torch.bfloat16)


@pytest.mark.parametrize("method", ["ocp", "nv"])
@pytest.mark.parametrize("fmt", list(FMTS))
@pytest.mark.parametrize("xname", ["f32", "bf16"])
def test_quantize_mx_oracle_and_host_formula_match_the_reference(xname, fmt, method):
    g = load("g7_quantize_mx.npz")
    x = _x(g, xname)
    want_q, want_s = g[f"q_{xname}_{fmt}_{method}"], g[f"s_{xname}_{fmt}_{method}"]
    q, s = mx_ref.quantize_mx(x.float().numpy(), fmt, method)
    assert np.array_equal(s, want_s)
    assert np.array_equal(q, want_q)
    qt, st = mx.quantize_mx(x, FMTS[fmt], method)              # host tensors: ATen formulas of the product package
    assert np.array_equal(st.view(torch.uint8).numpy(), want_s)
    assert np.array_equal(qt.contiguous().view(torch.uint8).numpy(), want_q)


@pytest.mark.parametrize("method", ["ocp", "nv"])
def test_dequantize_and_pack_match_the_reference(method):
    g = load("g7_quantize_mx.npz")
    q, s = g[f"q_f32_fp4_{method}"], g[f"s_f32_fp4_{method}"]
    want = g[f"deq_f32_fp4_{method}"]
    assert np.array_equal(mx_ref.dequantize_mxfp4(q, s).view(np.uint32), want.view(np.uint32))
    got = mx.dequantize_mxfp4(torch.from_numpy(q).view(mx.FP4), torch.from_numpy(s).view(mx.E8M0))
    assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(mx_ref.pack_block_scales_nv(g["pack_in"]), g["pack_out"])
    packed = mx.pack_block_scales_nv(torch.from_numpy(g["pack_in"]))
    assert np.array_equal(packed.numpy(), g["pack_out"])
    assert np.array_equal(mx.unpack_block_scales_nv(packed, 256, 8).numpy(), g["pack_in"])


def test_mx_ops_validate_like_the_reference():
    A = torch.zeros(4, 16, dtype=torch.uint8).view(mx.FP4)
    with pytest.raises(AssertionError):
        mx.mxfp4_mm(A, A, torch.zeros(4, 1, dtype=torch.uint8).view(mx.E8M0), torch.zeros(4, 1, dtype=torch.uint8).view(mx.E8M0))   # B.T not contiguous
    with pytest.raises(AssertionError):
        mx.quantize_mx(torch.zeros(4, 32), torch.int8)
    with pytest.raises(ValueError):
        mx.quantize_mx(torch.zeros(4, 32), mx.FP4, "bogus")
    out = torch.ops.gn_kernels.mxfp4_mm(A.to("meta"), A.to("meta").T.contiguous().T, torch.empty(4, 1, device="meta", dtype=mx.E8M0), torch.empty(16, 1, device="meta", dtype=mx.E8M0), None)
    assert out.shape == (4, 16) and out.dtype is torch.bfloat16


@pytest.mark.parametrize("tag", ["mfma_ocp", "mfma_nv", "generic_ocp", "generic_nv"])
def test_mxfp4_mm_oracle_matches_the_reference_derived_product(tag):
    """g10 (tools/gen_golden_mxfp4_mm.py): operands from the reference's quantize_mx, expectation = the reference's dequantize_mxfp4 of
    both, multiplied exactly.  The numpy oracle must reproduce it to the last bit (all factors are dyadic rationals: fp64 is exact)."""
    from oracle import mx_ref

    g = load("g10_mxfp4_mm.npz")
    want = g[f"C_{tag}"]
    got = mx_ref.mxfp4_mm(g[f"A_{tag}"], g[f"B_{tag}"], g[f"sA_{tag}"], g[f"sB_{tag}"])
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("tag", ["mfma_e4m3_e4m3", "mfma_e5m2_e4m3", "mfma_e4m3_e5m2", "generic_e4m3_e4m3"])
def test_mxfp8_mm_oracle_matches_the_reference_derived_product(tag):
    """g10: MXFP8 operands from the reference's quantize_mx (e4m3 / e5m2), expectation = exact product of code value x 2^(scale - 127)."""
    from oracle import mx_ref

    g = load("g10_mxfp4_mm.npz")
    fa, fb = tag.split("_")[1:]
    got = mx_ref.mxfp8_mm(g[f"A_f8_{tag}"], g[f"B_f8_{tag}"], g[f"sA_f8_{tag}"], g[f"sB_f8_{tag}"], fa, fb)
    assert np.array_equal(got, g[f"C_f8_{tag}"])
