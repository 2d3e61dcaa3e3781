"""Pins the CPU oracle (oracle/gnk_oracle.c and oracle/torch_ref.py) against golden vectors that
tools/gen_golden.py produced by importing the reference itself.  Integer and fp32 results are
compared bit-for-bit; fp8 products (hardware-defined accumulation order in the reference) use the
reference benchmark's own tolerances (benchmark_mm.py:128-145)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_ref
from tests._util import FLOAT_DTYPES, FP8_DTYPES, LAYOUTS, bits_equal, load, t_from_bits, to_layout


# ------------------------------------------------------------------------------- format helpers
def test_scalar_format_helpers_match_torch_casts():
    L = oracle.lib()
    g = torch.Generator().manual_seed(0)
    vals = torch.cat([torch.randn(2000, generator=g) * 10.0 ** torch.randint(-8, 8, (2000,), generator=g).float(),
                      torch.tensor([0.0, -0.0, 1.0, 65504.0, 65520.0, 1e-8, 6e-8, 3.0e38, float("inf"), -float("inf")])])
    for v in vals.tolist():
        f = torch.tensor(v, dtype=torch.float32)
        assert L.orc_f32_to_bf16(v) == (f.bfloat16().view(torch.int16).item() & 0xFFFF)
        assert L.orc_f32_to_f16(v) == (f.half().view(torch.int16).item() & 0xFFFF)
    for b in range(256):
        t = torch.tensor([b], dtype=torch.uint8)
        for name, fn in (("e4m3", L.orc_e4m3_to_f32), ("e5m2", L.orc_e5m2_to_f32)):
            want = t.view(FP8_DTYPES[name]).float().item()
            got = fn(b)
            assert (np.isnan(want) and np.isnan(got)) or want == got, (name, b, want, got)
    for h in range(0, 65536, 7):
        t = torch.tensor([h], dtype=torch.int32).to(torch.int16)
        assert L.orc_bf16_to_f32(h) == t.view(torch.bfloat16).float().item() or np.isnan(L.orc_bf16_to_f32(h))
        w = t.view(torch.float16).float().item()
        assert L.orc_f16_to_f32(h) == w or np.isnan(w)


# ------------------------------------------------------------------------------- G1 quantize
@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("impl", ["c", "aten"])
def test_quantize_int8_matches_reference(name, impl):
    g = load("g1_quantize_int8.npz")
    dt = FLOAT_DTYPES[name]
    x = t_from_bits(g[f"x_{name}"], dt)
    fn = oracle.quantize_int8 if impl == "c" else torch_ref.quantize_int8
    q, s = fn(x, dim=1)
    assert torch.equal(q, torch.from_numpy(g[f"q_{name}"]))
    assert bits_equal(s, t_from_bits(g[f"s_{name}"], dt).view(64, 1))
    # reduction along dim 0 and its transposed-view spelling keep x's strides
    q0, s0 = fn(x, dim=0)
    assert q0.stride() == x.stride()
    assert torch.equal(q0, torch.from_numpy(g[f"q_dim0_{name}"]))
    assert bits_equal(s0, t_from_bits(g[f"s_dim0_{name}"], dt).view(1, 256))
    qt, st = fn(x.T, dim=1)
    assert qt.stride() == x.T.stride() and tuple(st.shape) == (256, 1)
    assert torch.equal(qt.T, q0) and bits_equal(st.view(-1), s0.view(-1))


@pytest.mark.parametrize("impl", ["c", "aten"])
def test_quantize_int8_stochastic_rounding_replays_reference_draw(impl):
    g = load("g1_quantize_int8.npz")
    x = t_from_bits(g["x_bf16"], torch.bfloat16)
    noise = torch.from_numpy(g["sr_noise"])
    if impl == "c":
        q, s = oracle.quantize_int8(x, True, dim=1, noise=noise)
    else:
        q, s = torch_ref.quantize_int8(x, True, dim=1, noise=noise)
    assert torch.equal(q, torch.from_numpy(g["q_sr_bf16"]))
    assert bits_equal(s, t_from_bits(g["s_sr_bf16"], torch.bfloat16).view(64, 1))


def test_quantize_int8_eps():
    g = load("g1_quantize_int8.npz")
    x = t_from_bits(g["x_bf16"], torch.bfloat16)
    q, s = oracle.quantize_int8(x, dim=1, eps=1e-5)
    assert torch.equal(q, torch.from_numpy(g["q_eps1e5_bf16"]))
    assert bits_equal(s, t_from_bits(g["s_eps1e5_bf16"], torch.bfloat16).view(64, 1))


def test_philox_stream_is_uniform_and_keyed():
    u = np.array([oracle.sr_uniform(1, 0, i) for i in range(4000)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
    assert oracle.sr_uniform(1, 0, 5) == oracle.sr_uniform(1, 5, 0) != oracle.sr_uniform(2, 0, 5)


# ------------------------------------------------------------------------------- G2/G3 int8 mm
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_int8_mm_bit_exact(layout):
    g = load("g2_g3_int8_mm.npz")
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.from_numpy(g["A"]), acm)
    B = to_layout(torch.from_numpy(g["B"]), bcm)
    assert torch.equal(oracle.int8_mm(A, B), torch.from_numpy(g["C_i32"]))


@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_scaled_mm_int8_bit_exact(name, layout):
    g = load("g2_g3_int8_mm.npz")
    dt = FLOAT_DTYPES[name]
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.from_numpy(g["A"]), acm)
    B = to_layout(torch.from_numpy(g["B"]), bcm)
    rs, cs = t_from_bits(g[f"rs_{name}"], dt), t_from_bits(g[f"cs_{name}"], dt)
    want = torch.from_numpy(g[f"C_scaled_{name}"])
    assert bits_equal(oracle.scaled_mm(A, B, rs, cs, out_dtype=torch.float32), want)
    assert bits_equal(oracle.scaled_mm(A, B, rs, cs), want.to(dt))          # cast on store is RNE
    assert bits_equal(torch_ref.scaled_mm_exact(A, B, rs, cs, torch.float32), want)
    want_s = torch.from_numpy(g[f"C_scaled_scalar_{name}"])
    assert bits_equal(oracle.scaled_mm(A, B, rs, cs[:, :1].contiguous(), out_dtype=torch.float32), want_s)


def test_scaled_mm_meets_reference_benchmark_tolerance():
    """The reference's own check: scaled_mm vs dequantize-then-matmul (benchmark_mm.py:46-50,116-118)."""
    g = load("g2_g3_int8_mm.npz")
    A, B = torch.from_numpy(g["A"]), to_layout(torch.from_numpy(g["B"]), True)
    rs, cs = t_from_bits(g["rs_bf16"], torch.bfloat16), t_from_bits(g["cs_bf16"], torch.bfloat16)
    torch.testing.assert_close(oracle.scaled_mm(A, B, rs, cs), torch_ref.scaled_mm_dequant(A, B, rs, cs), atol=4e-4, rtol=1e-2)
    torch.testing.assert_close(torch_ref.scaled_mm_intmm(A, B, rs, cs), torch_ref.scaled_mm_dequant(A, B, rs, cs), atol=4e-4, rtol=1e-2)


# ------------------------------------------------------------------------------- G4 tile-scaled int8
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_tile_scaled_mm_int8_deepseek_blocks_bit_exact(layout):
    g = load("g4_tile_scaled_int8.npz")
    acm, bcm = LAYOUTS[layout]
    A, B = to_layout(torch.from_numpy(g["ds_A"]), acm), to_layout(torch.from_numpy(g["ds_B"]), bcm)
    sA = to_layout(t_from_bits(g["ds_sA"], torch.bfloat16), acm)
    sB = to_layout(t_from_bits(g["ds_sB"], torch.bfloat16), bcm)
    assert bits_equal(oracle.scaled_mm(A, B, sA, sB, out_dtype=torch.float32), torch.from_numpy(g["ds_C"]))


def test_tile_scaled_mm_int8_jetfire_blocks_bit_exact():
    g = load("g4_tile_scaled_int8.npz")
    A, B = torch.from_numpy(g["jf_A"]), to_layout(torch.from_numpy(g["jf_B"]), True)
    sA, sB = torch.from_numpy(g["jf_sA"]), torch.from_numpy(g["jf_sB"])
    assert bits_equal(oracle.scaled_mm(A, B, sA, sB), torch.from_numpy(g["jf_C"]))


# ------------------------------------------------------------------------------- G5 fp8
@pytest.mark.parametrize("tag", ["e4m3_e4m3", "e5m2_e5m2", "e4m3_e5m2"])
def test_fp8_products_within_reference_tolerance(tag):
    g = load("g5_fp8.npz")
    an, bn = tag.split("_")
    A = t_from_bits(g[f"A_{tag}"], FP8_DTYPES[an])
    B = to_layout(t_from_bits(g[f"B_{tag}"], FP8_DTYPES[bn]), True)
    rs, cs = t_from_bits(g["rs"], torch.bfloat16), t_from_bits(g["cs"], torch.bfloat16)
    sA, sB = t_from_bits(g["sA"], torch.bfloat16), t_from_bits(g["sB"], torch.bfloat16)
    # fp32 results: accumulation order differs from the interpreter's dot -> tight fp32 tolerance
    torch.testing.assert_close(oracle.scaled_mm(A, B, rs, cs, out_dtype=torch.float32), torch.from_numpy(g[f"C_scaled_{tag}"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(oracle.scaled_mm(A, B, sA, sB, out_dtype=torch.float32), torch.from_numpy(g[f"C_tile_{tag}"]), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(oracle.fp8_mm(A, B, out_dtype=torch.float32), torch.from_numpy(g[f"C_plain_{tag}"]), rtol=1e-4, atol=1e-4)
    # and the reference benchmark's bf16 check (default bf16 tolerances)
    torch.testing.assert_close(oracle.scaled_mm(A, B, rs, cs), torch.from_numpy(g[f"C_scaled_{tag}"]).bfloat16())
    torch.testing.assert_close(oracle.scaled_mm(A, B, rs, cs), torch_ref.scaled_mm_dequant(A, B, rs, cs))


def test_cutlass_epilogue_order_differs_only_in_last_ulp():
    g = load("g5_fp8.npz")
    A = t_from_bits(g["A_e4m3_e4m3"], torch.float8_e4m3fn)
    B = to_layout(t_from_bits(g["B_e4m3_e4m3"], torch.float8_e4m3fn), True)
    rs, cs = t_from_bits(g["rs"], torch.bfloat16).float(), t_from_bits(g["cs"], torch.bfloat16).float()
    a = oracle.scaled_mm(A, B, rs, cs, out_dtype=torch.float32, order=0)
    b = oracle.scaled_mm(A, B, rs, cs, out_dtype=torch.float32, order=1)
    torch.testing.assert_close(a, b, rtol=3e-7, atol=0)


@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("dim", [1, 0])
def test_quantize_nonfinite_golden(name, dim):
    """g9 (tools/gen_golden_nonfinite.py, from the reference on the CPU): a NaN makes its row's scale NaN, an Inf makes it Inf and every
    finite element 0.  Elements whose quotient is NaN are a cast of NaN in the reference (implementation-defined): compared outside the
    fixture's mask only; there this build writes 0."""
    import oracle
    from tests._util import FLOAT_DTYPES

    g = load("g9_quantize_nonfinite.npz")
    dt = FLOAT_DTYPES[name]
    x = t_from_bits(g[f"x_{name}"], dt)
    q, s = oracle.quantize_int8(x, dim=dim)
    want_s = t_from_bits(g[f"s_dim{dim}_{name}"], dt)
    assert torch.equal(torch.isnan(s.float()), torch.isnan(want_s.float()))
    fin = ~torch.isnan(want_s.float())
    assert torch.equal(s.float()[fin], want_s.float()[fin])
    undef = torch.from_numpy(g[f"undef_dim{dim}_{name}"])
    assert torch.equal(q[~undef], torch.from_numpy(g[f"q_dim{dim}_{name}"])[~undef])
    assert int(q[undef].abs().max()) == 0
