"""GPU parity of the MFMA GEMM kernels through the gn_kernels ops -> C ABI: int32 accumulators and
fp32 scaled outputs bit-exact against reference-generated goldens and the CPU oracle in all operand
layouts and kernel variants; fp8 within the reference benchmark's tolerances; full-size (4096^3)
checks through sampled rows/columns and epilogue-linearity properties."""
import pytest
import torch

import gn_kernels
import oracle
from gn_kernels import _cabi
from tests._util import FLOAT_DTYPES, FP8_DTYPES, LAYOUTS, bits_equal, load, t_from_bits, to_layout

pytestmark = pytest.mark.gpu
DEV = "cuda"
VARIANTS = {"auto": -1, "tile256": 0, "tile128": 1, "generic": 2, "tile256simple": 3, "tile256pp8": 4, "tile256w4": 5, "tile64x128": 6}


@pytest.fixture(params=list(VARIANTS))
def variant(request):
    L = _cabi.load()
    L.gnk_set_gemm_variant(VARIANTS[request.param])
    yield request.param
    L.gnk_set_gemm_variant(-1)


def dev_layout(x, column_major):
    return to_layout(x.to(DEV), column_major)


class tile_rescale:
    """`with tile_rescale("exact"):` -- the three-rounding rescale of the reference's interpreter for every output dtype (the default uses the
    fused multiply-add form for bf16 / fp16 outputs: include/gnk.h, gnk_set_tile_rescale)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from gn_kernels import mm
        self.old = mm.set_tile_rescale(self.mode)

    def __exit__(self, *exc):
        from gn_kernels import mm
        mm.set_tile_rescale(self.old)


def assert_tile_int8_close(got, want32):
    """The reference's own check of tile-scaled INT8 (benchmark_mm.py:119-121): atol 1e-4, rtol 1e-2 against the fp32 result rounded to the
    output dtype."""
    torch.testing.assert_close(got.float().cpu(), want32.to(got.dtype).float().cpu(), atol=1e-4, rtol=1e-2)


def fp8_mag(A, B, sa=None, sb=None):
    """sum_k |a_mk * b_kn| (times |scales|): the magnitude the fp8 MFMA's accumulation error scales with."""
    a, b = A.float().abs().cpu(), B.float().abs().cpu()
    if sa is not None:
        M, K = a.shape
        N = b.shape[1]
        sa, sb = sa.float().abs().cpu(), sb.float().abs().cpu()
        a = a * sa.repeat_interleave(M // sa.shape[0], 0).repeat_interleave(K // sa.shape[1], 1)
        b = b * sb.repeat_interleave(K // sb.shape[0], 0).repeat_interleave(N // sb.shape[1], 1)
    return a @ b


def assert_fp8_close(got, want, mag):
    """STATED TOLERANCE for fp8 products on MI355X:  |got - want| <= rtol*|want| + 5e-5 * sum_k|a*b| + 1e-5,
    rtol = 1.6e-2 for bf16/fp16 outputs (the reference's assert_close default, benchmark_mm.py:128-145)
    and 2e-3 for fp32 outputs.  The middle term is a hardware property: CDNA4's fp8 MFMAs (both
    v_mfma_f32_16x16x32_fp8_fp8 and v_mfma_scale_f32_16x16x128_f8f6f4) sum their products with less
    than fp32 precision -- measured max |err| / sum|a*b| = 1.4e-5 at K=128 on randn data
    (tools/fp8_err.py), identical for both instructions -- so an output that nearly cancels can miss
    a purely relative bound that NVIDIA's fp32-accumulating MMA meets."""
    rtol = 2e-3 if got.dtype is torch.float32 else 1.6e-2
    got, want = got.float().cpu(), want.float().cpu()
    bound = rtol * want.abs() + 5e-5 * mag + 1e-5
    bad = (got - want).abs() > bound
    assert not bool(bad.any()), f"{int(bad.sum())} elements outside the fp8 tolerance; worst excess {((got - want).abs() - bound).max().item():.3e}"


# ------------------------------------------------------------------------------- goldens
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_int8_mm_golden_bit_exact(layout, variant):
    g = load("g2_g3_int8_mm.npz")
    acm, bcm = LAYOUTS[layout]
    A, B = dev_layout(torch.from_numpy(g["A"]), acm), dev_layout(torch.from_numpy(g["B"]), bcm)
    C = gn_kernels.int8_mm(A, B)
    assert C.dtype is torch.int32 and torch.equal(C.cpu(), torch.from_numpy(g["C_i32"]))


@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_scaled_mm_int8_golden_bit_exact(layout, name, variant):
    g = load("g2_g3_int8_mm.npz")
    dt = FLOAT_DTYPES[name]
    acm, bcm = LAYOUTS[layout]
    A, B = dev_layout(torch.from_numpy(g["A"]), acm), dev_layout(torch.from_numpy(g["B"]), bcm)
    rs, cs = t_from_bits(g[f"rs_{name}"], dt).to(DEV), t_from_bits(g[f"cs_{name}"], dt).to(DEV)
    want32 = torch.from_numpy(g[f"C_scaled_{name}"])
    out = gn_kernels.scaled_mm(A, B, rs, cs)
    assert out.dtype is dt and bits_equal(out, want32.to(dt))          # kernel result == RNE(fp32 golden)
    if name == "f32":
        assert bits_equal(out, want32)
    outs = gn_kernels.scaled_mm(A, B, rs, cs[:, :1].contiguous())         # COL_SCALE_SCALAR
    assert bits_equal(outs, torch.from_numpy(g[f"C_scaled_scalar_{name}"]).to(dt))


@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_tile_scaled_int8_deepseek_golden_bit_exact(layout, variant):
    if variant in ("tile256", "tile256simple", "tile256pp8", "tile256w4"):
        pytest.skip("tile-scaled epilogue has no 256x256 variant (uses the 128x128 tile)")
    g = load("g4_tile_scaled_int8.npz")
    acm, bcm = LAYOUTS[layout]
    A, B = dev_layout(torch.from_numpy(g["ds_A"]), acm), dev_layout(torch.from_numpy(g["ds_B"]), bcm)
    sA = dev_layout(t_from_bits(g["ds_sA"], torch.bfloat16), acm).float()
    sB = dev_layout(t_from_bits(g["ds_sB"], torch.bfloat16), bcm).float()
    assert bits_equal(gn_kernels.scaled_mm(A, B, sA, sB), torch.from_numpy(g["ds_C"]))      # fp32 scales -> fp32 out, exact
    sAb, sBb = sA.bfloat16(), sB.bfloat16()
    with tile_rescale("exact"):
        assert bits_equal(gn_kernels.scaled_mm(A, B, sAb, sBb), torch.from_numpy(g["ds_C"]).bfloat16())
    # default for 16-bit outputs: the fused form, held to the reference's tolerance for this op; forced onto the fp32 output too
    assert_tile_int8_close(gn_kernels.scaled_mm(A, B, sAb, sBb), torch.from_numpy(g["ds_C"]))
    if variant != "generic":
        with tile_rescale("fused"):
            got = gn_kernels.scaled_mm(A, B, sA, sB)
        assert_tile_int8_close(got, torch.from_numpy(g["ds_C"]))
        # ... and it is a different rounding sequence, not the same kernel under another name: some fp32 outputs differ in the last ulps
        want = torch.from_numpy(g["ds_C"])
        assert not bits_equal(got, want) and (got.cpu() - want).abs().max() <= 2e-6 * want.abs().max()


def test_tile_scaled_int8_jetfire_golden_bit_exact():
    g = load("g4_tile_scaled_int8.npz")
    A, B = torch.from_numpy(g["jf_A"]).to(DEV), dev_layout(torch.from_numpy(g["jf_B"]), True)
    sA, sB = torch.from_numpy(g["jf_sA"]).to(DEV), torch.from_numpy(g["jf_sB"]).to(DEV)
    assert bits_equal(gn_kernels.scaled_mm(A, B, sA, sB), torch.from_numpy(g["jf_C"]))


@pytest.mark.parametrize("tag", ["e4m3_e4m3", "e5m2_e5m2", "e4m3_e5m2"])
def test_fp8_goldens_within_reference_tolerance(tag, variant):
    g = load("g5_fp8.npz")
    an, bn = tag.split("_")
    A = t_from_bits(g[f"A_{tag}"], FP8_DTYPES[an]).to(DEV)
    B = dev_layout(t_from_bits(g[f"B_{tag}"], FP8_DTYPES[bn]), True)
    rs, cs = t_from_bits(g["rs"], torch.bfloat16).to(DEV), t_from_bits(g["cs"], torch.bfloat16).to(DEV)
    sA, sB = t_from_bits(g["sA"], torch.bfloat16).to(DEV), t_from_bits(g["sB"], torch.bfloat16).to(DEV)
    C_scaled, C_tile, plain = (torch.from_numpy(g[f"C_{k}_{tag}"]) for k in ("scaled", "tile", "plain"))
    mag, mag_s, mag_t = fp8_mag(A, B), fp8_mag(A, B, rs, cs), fp8_mag(A, B, sA, sB)
    assert_fp8_close(gn_kernels.scaled_mm(A, B, rs, cs), C_scaled, mag_s)                       # bf16 out (benchmark_mm.py:142)
    assert_fp8_close(gn_kernels.scaled_mm(A, B, rs.float(), cs.float()), C_scaled, mag_s)       # fp32 out
    if variant not in ("tile256", "tile256simple", "tile256pp8", "tile256w4"):
        assert_fp8_close(gn_kernels.scaled_mm(A, B, sA, sB), C_tile, mag_t)
        assert_fp8_close(gn_kernels.scaled_mm(A, B, sA.float(), sB.float()), C_tile, mag_t)
    assert_fp8_close(gn_kernels._triton_mm(A, B, out_dtype=torch.bfloat16), plain, mag)         # benchmark_mm.py:129
    if tag == "e4m3_e4m3":
        assert_fp8_close(gn_kernels.fp8_mm(A, B), plain, mag)
        # CUTLASS-named op: fp32 scales, (acc*col)*row, bf16 out (benchmark_mm.py:135-137)
        out = gn_kernels.scaled_fp8_mm(A, B, rs.float(), cs.float())
        assert out.dtype is torch.bfloat16
        assert_fp8_close(out, C_scaled, mag_s)


# ------------------------------------------------------------------------------- seeded random vs oracle
@pytest.mark.parametrize("shape", [(256, 256, 128), (300, 200, 384), (512, 384, 1024), (130, 70, 256), (1, 16, 128), (100, 52, 200), (33, 17, 9)])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_int8_paths_vs_oracle(shape, layout):
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 1000 + N)
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g), acm)
    B = to_layout(torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g), bcm)
    rs = torch.randn(M, 1, generator=g).div(128).bfloat16()
    cs = torch.randn(1, N, generator=g).div(128).bfloat16()
    Ad, Bd = dev_layout(A, acm), dev_layout(B, bcm)
    assert Ad.stride() == A.stride() and Bd.stride() == B.stride()
    assert torch.equal(gn_kernels.int8_mm(Ad, Bd).cpu(), oracle.int8_mm(A, B))
    assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, rs.to(DEV), cs.to(DEV)), oracle.scaled_mm(A, B, rs, cs))
    assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, rs.float().to(DEV), cs.float().to(DEV)), oracle.scaled_mm(A, B, rs.float(), cs.float()))


@pytest.mark.parametrize("blocks", [(1, 128, 128), (128, 128, 128), (1, 1, 128), (32, 32, 32), (16, 64, 64), (4, 8, 16)])
def test_tile_scaled_int8_vs_oracle(blocks):
    QM, QN, QK = blocks
    M, N, K = 128, 256, 384
    g = torch.Generator().manual_seed(QM + QN)
    A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g)
    B = to_layout(torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g), True)
    sA = torch.randn(M // QM, K // QK, generator=g).div(128)
    sB = torch.randn(K // QK, N // QN, generator=g).div(128)
    got = gn_kernels.scaled_mm(A.to(DEV), dev_layout(B, True), sA.to(DEV), sB.to(DEV))
    assert bits_equal(got, oracle.scaled_mm(A, B, sA, sB))


@pytest.mark.parametrize("qmn", [(1, 1), (128, 2), (1, 3), (4, 6)])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_tile_scaled_per_column_scale_blocks(qmn, layout):
    """Scale blocks whose width QN is not a multiple of 4 (per-column scales and the like; the reference takes any block size its shapes
    give, triton_mm.py:419,434-437): 128-k blocks run on MFMA-kernel instances that read one B-side scale per COLUMN from the LDS panel
    (GemmCfg::PERCOL).  int8 with fp32 scales bit for bit against the oracle (exact rescale), bf16 scales in the default fused form at the
    reference's tolerance, fp8 at the stated fp8 tolerance; ragged M / N edges, K over several scale panels."""
    QM, QN = qmn
    QK = 128
    M, N, K = 384 if QM == 128 else 400, 528, 128 * 11      # multiples of 16 (M/N-major operands), ragged against the 128 x 128 tile
    assert M % QM == 0 and N % QN == 0 and N % 4 == 0 and QN % 4 != 0
    g = torch.Generator().manual_seed(QM * 131 + QN * 17 + len(layout))
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g), acm)
    B = to_layout(torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g), bcm)
    sA = torch.randn(M // QM, K // QK, generator=g).div(128)
    sB = torch.randn(K // QK, N // QN, generator=g).div(128)
    Ad, Bd = dev_layout(A, acm), dev_layout(B, bcm)
    L = _cabi.load()
    L.gnk_set_gemm_variant(VARIANTS["tile128"])            # a forced fast variant errors out instead of falling back to the generic kernel
    try:
        got = gn_kernels.scaled_mm(Ad, Bd, sA.to(DEV), sB.to(DEV))
        assert bits_equal(got, oracle.scaled_mm(A, B, sA, sB))
        want32 = oracle.scaled_mm(A, B, sA.bfloat16().float(), sB.bfloat16().float())
        assert_tile_int8_close(gn_kernels.scaled_mm(Ad, Bd, sA.bfloat16().to(DEV), sB.bfloat16().to(DEV)), want32)
        with tile_rescale("exact"):
            assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, sA.bfloat16().to(DEV), sB.bfloat16().to(DEV)), want32.bfloat16())
        A8 = to_layout((torch.randn(M, K, generator=g) * 0.5).to(torch.float8_e4m3fn), acm)
        B8 = to_layout((torch.randn(K, N, generator=g) * 0.5).to(torch.float8_e4m3fn), bcm)
        got8 = gn_kernels.scaled_mm(dev_layout(A8, acm), dev_layout(B8, bcm), sA.to(DEV), sB.to(DEV))
        assert_fp8_close(got8, oracle.scaled_mm(A8, B8, sA, sB), fp8_mag(A8, B8, sA, sB))
    finally:
        L.gnk_set_gemm_variant(-1)


@pytest.mark.parametrize("blocks", [(32, 32, 32), (64, 64, 64), (1, 128, 32), (32, 4, 64), (128, 128, 256), (1, 64, 640), (32, 32, 1280)])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_tile_scaled_small_k_blocks_int8_bit_exact(blocks, layout):
    """Jetfire-style blocks (BASELINE configs[2]): 64- and 32-k scale blocks on the 32-k MFMA instances, and blocks of a multiple of
    128 k (partial sums carried over several K-steps) -- every operand layout, bf16 and fp32 scales, K longer than one scale panel
    (8 scale steps), ragged M / N edges."""
    QM, QN, QK = blocks
    M, N, K = 384 + QM * 2, 512 + QN * 3 if QN >= 4 else 512, 1280
    M, N = (M // QM) * QM, (N // QN) * QN
    g = torch.Generator().manual_seed(QM * 7 + QN + QK)
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g), acm)
    B = to_layout(torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g), bcm)
    sA = torch.randn(M // QM, K // QK, generator=g).div(128)
    sB = torch.randn(K // QK, N // QN, generator=g).div(128)
    Ad, Bd = dev_layout(A, acm), dev_layout(B, bcm)
    assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, sA.to(DEV), sB.to(DEV)), oracle.scaled_mm(A, B, sA, sB))
    sAh, sBh = sA.bfloat16(), sB.bfloat16()
    want_h = oracle.scaled_mm(A, B, sAh, sBh)
    with tile_rescale("exact"):
        assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, sAh.to(DEV), sBh.to(DEV)), want_h)
    torch.testing.assert_close(gn_kernels.scaled_mm(Ad, Bd, sAh.to(DEV), sBh.to(DEV)).float().cpu(), want_h.float(), atol=1e-4, rtol=1e-2)   # fused (default)
    with tile_rescale("fused"):
        torch.testing.assert_close(gn_kernels.scaled_mm(Ad, Bd, sA.to(DEV), sB.to(DEV)).cpu(), oracle.scaled_mm(A, B, sA, sB), atol=1e-4, rtol=1e-2)
    sAt = sA.T.contiguous().T                                                    # column-major sA, as benchmark_mm.py:91-96 lays it out
    assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, sAt.to(DEV).T.contiguous().T, sB.to(DEV)), oracle.scaled_mm(A, B, sA, sB))


@pytest.mark.parametrize("fmts", [("e4m3", "e4m3"), ("e5m2", "e4m3"), ("e4m3", "e5m2"), ("e5m2", "e5m2")])
@pytest.mark.parametrize("qk", [32, 64])
def test_tile_scaled_small_k_blocks_fp8(fmts, qk):
    M, N, K = 256, 384, 640
    g = torch.Generator().manual_seed(qk + len(fmts[0]))
    for layout, (acm, bcm) in LAYOUTS.items():
        A = to_layout(torch.randn(M, K, generator=g).to(FP8_DTYPES[fmts[0]]), acm)
        B = to_layout(torch.randn(K, N, generator=g).to(FP8_DTYPES[fmts[1]]), bcm)
        sA = torch.randn(M // 32, K // qk, generator=g).div(128)
        sB = torch.randn(K // qk, N // 32, generator=g).div(128)
        got = gn_kernels.scaled_mm(dev_layout(A, acm), dev_layout(B, bcm), sA.to(DEV), sB.to(DEV))
        assert_fp8_close(got, oracle.scaled_mm(A, B, sA, sB), fp8_mag(A, B, sA, sB))


@pytest.mark.parametrize("fmts", [("e4m3", "e4m3"), ("e5m2", "e4m3"), ("e4m3", "e5m2"), ("e5m2", "e5m2")])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_fp8_paths_vs_oracle(fmts, layout, variant):
    if variant == "tile256simple":
        pytest.skip("int8-only variant")
    M, N, K = 256, 384, 512
    g = torch.Generator().manual_seed(11)
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randn(M, K, generator=g).to(FP8_DTYPES[fmts[0]]), acm)
    B = to_layout(torch.randn(K, N, generator=g).to(FP8_DTYPES[fmts[1]]), bcm)
    rs = torch.randn(M, 1, generator=g).div(128)
    cs = torch.randn(1, N, generator=g).div(128)
    Ad, Bd = dev_layout(A, acm), dev_layout(B, bcm)
    assert_fp8_close(gn_kernels.scaled_mm(Ad, Bd, rs.to(DEV), cs.to(DEV)), oracle.scaled_mm(A, B, rs, cs), fp8_mag(A, B, rs, cs))
    assert_fp8_close(gn_kernels._triton_mm(Ad, Bd, out_dtype=torch.float32), oracle.fp8_mm(A, B, out_dtype=torch.float32), fp8_mag(A, B))
    assert_fp8_close(gn_kernels.scaled_mm(Ad, Bd, rs.bfloat16().to(DEV), cs.bfloat16().to(DEV)),
                     oracle.scaled_mm(A, B, rs.bfloat16(), cs.bfloat16(), out_dtype=torch.float32), fp8_mag(A, B, rs, cs))


@pytest.mark.parametrize("shape", [(256, 256, 64), (300, 520, 192), (512, 256, 320), (1, 8, 64), (257, 263, 448)])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_four_stage_kernel_k64_multiples_and_ragged_edges(shape, layout):
    """The 256 x 256 four-stage kernel forced onto K = 64 * odd, fewer stages than its pipeline depth and
    ragged M / N (clamped staging rows, masked stores, scalar / vector / LDS-staged epilogues)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    acm, bcm = LAYOUTS[layout]
    if (acm and M % 16) or ((not bcm) and N % 16):
        pytest.skip("fast path needs 16-byte aligned leading dimensions")
    A = to_layout(torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g), acm)
    B = to_layout(torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g), bcm)
    rs = torch.randn(M, 1, generator=g).div(128).bfloat16()
    cs = torch.randn(1, N, generator=g).div(128).bfloat16()
    A8 = to_layout(torch.randn(M, K, generator=g).to(torch.float8_e4m3fn), acm)
    B8 = to_layout(torch.randn(K, N, generator=g).to(torch.float8_e5m2), bcm)
    L = _cabi.load()
    L.gnk_set_gemm_variant(0)
    try:
        Ad, Bd = dev_layout(A, acm), dev_layout(B, bcm)
        assert torch.equal(gn_kernels.int8_mm(Ad, Bd).cpu(), oracle.int8_mm(A, B))
        assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, rs.to(DEV), cs.to(DEV)), oracle.scaled_mm(A, B, rs, cs))
        assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, rs.float().to(DEV), cs.float().to(DEV)), oracle.scaled_mm(A, B, rs.float(), cs.float()))
        got8 = gn_kernels.scaled_mm(dev_layout(A8, acm), dev_layout(B8, bcm), rs.float().to(DEV), cs.float().to(DEV))
        assert_fp8_close(got8, oracle.scaled_mm(A8, B8, rs.float(), cs.float()), fp8_mag(A8, B8, rs, cs))
    finally:
        L.gnk_set_gemm_variant(-1)


@pytest.mark.parametrize("shape", [(1024, 1024, 8192), (384, 640, 4096), (1024, 2048, 8192), (130, 260, 4224)])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_split_k_is_bit_identical(shape, layout):
    """Long K with few output tiles (the grad-weight GEMM shapes): the op offers a workspace, the library splits K,
    and the result must equal the oracle -- and the unsplit kernel -- bit for bit (integer partial sums)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    acm, bcm = LAYOUTS[layout]
    if (acm and M % 16) or ((not bcm) and N % 16):
        pytest.skip("fast path needs 16-byte aligned leading dimensions")
    A = to_layout(torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g), acm)
    B = to_layout(torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g), bcm)
    rs = torch.randn(M, 1, generator=g).div(128).bfloat16()
    cs = torch.randn(1, N, generator=g).div(128).bfloat16()
    L = _cabi.load()
    want_ws = L.gnk_scaled_mm_workspace_bytes(_cabi.I8, _cabi.I8, M, N, K)
    assert want_ws > 0 and want_ws % (M * N * 4) == 0, "these shapes are meant to take the split-K path"
    Ad, Bd = dev_layout(A, acm), dev_layout(B, bcm)
    for r, c in ((rs, cs), (rs.float(), cs.float()), (rs.half(), cs[:, :1].half().contiguous())):
        got = gn_kernels.scaled_mm(Ad, Bd, r.to(DEV), c.to(DEV))
        assert bits_equal(got, oracle.scaled_mm(A, B, r, c))
        L.gnk_set_gemm_variant(1)            # forced variant = never split
        try:
            assert bits_equal(got, gn_kernels.scaled_mm(Ad, Bd, r.to(DEV), c.to(DEV)))
        finally:
            L.gnk_set_gemm_variant(-1)


def test_degenerate_shapes():
    """K = 0 (empty reduction: all zeros), a single row / column, and a row-scaled product whose scale is zero."""
    A0 = torch.zeros(5, 0, dtype=torch.int8, device=DEV)
    B0 = torch.zeros(0, 7, dtype=torch.int8, device=DEV)
    assert torch.equal(gn_kernels.int8_mm(A0, B0).cpu(), torch.zeros(5, 7, dtype=torch.int32))
    rs, cs = torch.ones(5, 1, device=DEV).bfloat16(), torch.ones(1, 7, device=DEV).bfloat16()
    assert torch.equal(gn_kernels.scaled_mm(A0, B0, rs, cs).float().cpu(), torch.zeros(5, 7))
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-128, 128, (1, 256), dtype=torch.int8, generator=g)
    B = to_layout(torch.randint(-128, 128, (256, 1), dtype=torch.int8, generator=g), True)
    rs1, cs1 = torch.randn(1, 1, generator=g).bfloat16(), torch.randn(1, 1, generator=g).bfloat16()
    assert bits_equal(gn_kernels.scaled_mm(A.to(DEV), dev_layout(B, True), rs1.to(DEV), cs1.to(DEV)), oracle.scaled_mm(A, B, rs1, cs1))
    A = torch.randint(-128, 128, (256, 128), dtype=torch.int8, generator=g)
    B = to_layout(torch.randint(-128, 128, (128, 256), dtype=torch.int8, generator=g), True)
    rz = torch.zeros(256, 1).bfloat16()
    cz = torch.randn(1, 256, generator=g).bfloat16()
    out = gn_kernels.scaled_mm(A.to(DEV), dev_layout(B, True), rz.to(DEV), cz.to(DEV))
    assert bits_equal(out, oracle.scaled_mm(A, B, rz, cz))     # signed zeros included


def test_errors_surface_as_exceptions():
    A = torch.zeros(4, 8, dtype=torch.int8, device=DEV)
    with pytest.raises(RuntimeError):
        torch.ops.gn_kernels.scaled_mm(A, A.T.contiguous(), torch.ones(4, 1, device=DEV, dtype=torch.float64), torch.ones(1, 4, device=DEV, dtype=torch.float64))
    L = _cabi.load()
    L.gnk_set_gemm_variant(0)
    try:
        with pytest.raises(RuntimeError, match="forced fast variant"):
            gn_kernels.int8_mm(torch.zeros(33, 9, dtype=torch.int8, device=DEV), torch.zeros(9, 17, dtype=torch.int8, device=DEV))
    finally:
        L.gnk_set_gemm_variant(-1)
    assert gn_kernels.int8_mm(torch.zeros(0, 8, dtype=torch.int8, device=DEV), torch.zeros(8, 4, dtype=torch.int8, device=DEV)).shape == (0, 4)


# ------------------------------------------------------------------------------- benchmark size
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_full_size_4096_int8(layout):
    """benchmark_mm.py inputs at 4096^3: sampled rows x columns against the oracle (bit-exact) and the
    fused epilogue against the unfused int32 result (bit-exact), per layout."""
    torch.manual_seed(2025 * 2024)
    S = 4096
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), acm)
    B = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), bcm)
    rs = torch.randn(S, 1, device=DEV).div(128).bfloat16()
    cs = torch.randn(1, S, device=DEV).div(128).bfloat16()
    C = gn_kernels.int8_mm(A, B)
    rows = torch.randperm(S)[:48].sort().values
    cols = torch.randperm(S)[:48].sort().values
    want = oracle.int8_mm(A[rows.to(DEV)].cpu().contiguous(), B[:, cols.to(DEV)].cpu().contiguous())
    assert torch.equal(C[rows.to(DEV)][:, cols.to(DEV)].cpu(), want)
    # checksum of checksums: column sums of C == (column sums of A) @ B in exact int64 arithmetic
    colsum = C.sum(0, dtype=torch.int64)
    want_colsum = (A.sum(0, dtype=torch.int64).double() @ B.double()).to(torch.int64)       # |values| < 2^53: exact in fp64
    assert torch.equal(colsum, want_colsum)
    out = gn_kernels.scaled_mm(A, B, rs, cs)
    assert bits_equal(out, ((C.float() * rs.float()) * cs.float()).bfloat16())
    torch.testing.assert_close(out[:64].float(), ((A[:64].float() * rs[:64].float()) @ (B.float() * cs.float())), atol=4e-4, rtol=1e-2)


def test_full_size_4096_tile_scaled_and_fp8():
    torch.manual_seed(2025 * 2024)
    S = 4096
    A = torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV)
    B = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), True)
    tsA = torch.randn(S, S // 128, device=DEV).div(128).bfloat16()
    tsB = to_layout(torch.randn(S // 128, S // 128, device=DEV).div(128).bfloat16(), True)
    with tile_rescale("exact"):
        out = gn_kernels.scaled_mm(A, B, tsA, tsB)
    rows = torch.arange(0, S, 97, device=DEV)
    want = oracle.scaled_mm(A[rows].cpu().contiguous(), B[:, :256].cpu(), tsA[rows].cpu().contiguous(), tsB[:, :2].cpu())
    assert bits_equal(out[rows][:, :256], want)
    # the default (fused) form for bf16 outputs: the reference's check of this op (benchmark_mm.py:119-121) against scaled_mm_ref, everywhere
    outf = gn_kernels.scaled_mm(A, B, tsA, tsB)
    ref = _scaled_ref(A, B, tsA, tsB).bfloat16()
    torch.testing.assert_close(outf, ref, atol=1e-4, rtol=1e-2)
    torch.testing.assert_close(outf[rows][:, :256].cpu(), want, atol=1e-4, rtol=1e-2)
    assert (outf != out).float().mean().item() < 0.05      # the two forms differ by a bf16 ulp in a few percent of the outputs at most
    A8 = torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn)
    B8 = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), True)
    rs = torch.randn(S, 1, device=DEV).div(128).bfloat16()
    cs = torch.randn(1, S, device=DEV).div(128).bfloat16()
    out8 = gn_kernels.scaled_mm(A8, B8, rs, cs)
    want8 = (A8[:128].float() * rs[:128].float()) @ (B8.float() * cs.float())                   # scaled_mm_ref, benchmark_mm.py:46-50
    assert_fp8_close(out8[:128], want8, fp8_mag(A8[:128], B8, rs[:128], cs))


@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_full_size_4096_jetfire_blocks(layout):
    """BASELINE configs[2] at its full size: 32 x 32 x 32 scale blocks.  Row blocks of the output depend on the same rows of A
    and of sA only, so four 32-row blocks spread over the matrix are checked bit-for-bit against the oracle over all of N and K;
    linearity in the scales (doubling sA doubles every fp32 output exactly) covers the rest."""
    torch.manual_seed(2025 * 2024 + 1)
    S, Q = 4096, 32
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), acm)
    B = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), bcm)
    sA = torch.randn(S // Q, S // Q, device=DEV).div(128)
    sB = torch.randn(S // Q, S // Q, device=DEV).div(128)
    out = gn_kernels.scaled_mm(A, B, sA, sB)
    assert out.dtype is torch.float32 and out.shape == (S, S)
    Bc = B.cpu()
    for blk in (0, 41, 86, 127):
        r = slice(blk * Q, (blk + 1) * Q)
        want = oracle.scaled_mm(A[r].cpu().contiguous(), Bc, sA[blk:blk + 1].cpu().contiguous(), sB.cpu())
        assert bits_equal(out[r], want)
    assert torch.equal(gn_kernels.scaled_mm(A, B, sA * 2, sB), out * 2)        # power-of-two scaling commutes with every rounding
    outh = gn_kernels.scaled_mm(A, B, sA.bfloat16(), sB.bfloat16())
    assert outh.dtype is torch.bfloat16 and torch.isfinite(outh.float()).all()
    torch.testing.assert_close(outh, _scaled_ref(A, B, sA.bfloat16(), sB.bfloat16()).bfloat16(), atol=1e-4, rtol=1e-2)   # fused form, benchmark_mm.py:119-121


# ------------------------------------------------------------------------------- round-2 parity gaps
def _scaled_ref(A, B, sa, sb):
    """scaled_mm_ref of benchmark_mm.py:46-50 in fp32 on the device (row / column / 2-D tile scales alike)."""
    M, K = A.shape
    N = B.shape[1]
    a = A.float() * sa.float().repeat_interleave(M // sa.shape[0], 0).repeat_interleave(K // sa.shape[1], 1) if sa.shape[1] > 1 else A.float() * sa.float()
    b = B.float() * sb.float().repeat_interleave(K // sb.shape[0], 0).repeat_interleave(N // sb.shape[1], 1) if sb.shape[0] > 1 else B.float() * sb.float()
    return a @ b


def _unwidened_failures(got, want32):
    """Elements the REFERENCE's own check would reject: torch.testing.assert_close(bf16, bf16) defaults, rtol 1.6e-2, atol 1e-5
    (benchmark_mm.py:128-145), against the bf16-rounded fp32 product."""
    want = want32.bfloat16().float()
    return int(((got.float() - want).abs() > 1.6e-2 * want.abs() + 1e-5).sum().item())


def _record(name, value):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(path):
        f = os.path.join(path, "fp8_unwidened_counts.json")
        data = json.load(open(f)) if os.path.exists(f) else {}
        data[name] = value
        json.dump(data, open(f, "w"), indent=1, sort_keys=True)
    print(f"[fp8 unwidened assert_close] {name}: {value}")


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_full_size_4096_fp8_scaled_all_layouts(layout):
    """benchmark_mm.py's FP8 inputs at 4096^3 in A@B.T, A@B and A.T@B: every output against scaled_mm_ref (fp32 on the device) under the
    stated tolerance, and the COUNT of elements the reference's unwidened assert_close would reject (reported)."""
    torch.manual_seed(2025 * 2024)
    S = 4096
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), acm)
    B = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), bcm)
    rs = torch.randn(S, 1, device=DEV).div(128).bfloat16()
    cs = torch.randn(1, S, device=DEV).div(128).bfloat16()
    out = gn_kernels.scaled_mm(A, B, rs, cs)
    want = _scaled_ref(A, B, rs, cs)
    mag = (A.float().abs() * rs.float().abs()) @ (B.float().abs() * cs.float().abs())
    bound = 1.6e-2 * want.abs() + 5e-5 * mag + 1e-5
    assert not bool(((out.float() - want).abs() > bound).any())
    failing = _unwidened_failures(out, want)
    _record(f"scaled_fp8_4096_{layout}", {"failing": failing, "of": S * S})
    assert failing == 0, "scaled fp8 must pass the reference's own unwidened assert_close (benchmark_mm.py:142-143)"
    plain = gn_kernels._triton_mm(A, B, out_dtype=torch.bfloat16)
    wantp = A.float() @ B.float()
    magp = A.float().abs() @ B.float().abs()
    assert not bool(((plain.float() - wantp).abs() > 1.6e-2 * wantp.abs() + 5e-5 * magp + 1e-5).any())
    _record(f"plain_fp8_4096_{layout}", {"failing": _unwidened_failures(plain, wantp), "of": S * S})


@pytest.mark.parametrize("style", ["deepseek", "jetfire"])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_full_size_4096_fp8_tile_scaled(style, layout):
    """Tile-scaled FP8 at 4096^3, DeepSeek-style (1 x 128 / 128 x 128) and Jetfire-style (32^3) blocks: whole output against
    scaled_mm_ref under the stated fp8 tolerance; sampled row blocks against the CPU oracle's ascending-K-block accumulation."""
    torch.manual_seed(2025 * 2024 + 3)
    S = 4096
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), acm)
    B = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), bcm)
    if style == "deepseek":
        sA = to_layout(torch.randn(S, S // 128, device=DEV).div(128).bfloat16(), acm)      # benchmark_mm.py:88-96
        sB = to_layout(torch.randn(S // 128, S // 128, device=DEV).div(128).bfloat16(), bcm)
        qm = 1
    else:
        sA = torch.randn(S // 32, S // 32, device=DEV).div(128).bfloat16()
        sB = torch.randn(S // 32, S // 32, device=DEV).div(128).bfloat16()
        qm = 32
    out = gn_kernels.scaled_mm(A, B, sA, sB)
    assert out.dtype is torch.bfloat16
    want = _scaled_ref(A, B, sA, sB)
    K = S
    a_abs = A.float().abs() * sA.float().abs().repeat_interleave(S // sA.shape[0], 0).repeat_interleave(K // sA.shape[1], 1)
    b_abs = B.float().abs() * sB.float().abs().repeat_interleave(K // sB.shape[0], 0).repeat_interleave(S // sB.shape[1], 1)
    mag = a_abs @ b_abs
    bound = 1.6e-2 * want.abs() + 5e-5 * mag + 1e-5
    assert not bool(((out.float() - want).abs() > bound).any())
    failing = _unwidened_failures(out, want)
    _record(f"tile_fp8_4096_{style}_{layout}", {"failing": failing, "of": S * S})
    assert failing == 0, "tile-scaled fp8 must pass the reference's own unwidened assert_close (benchmark_mm.py:144-145)"
    # oracle on one block of rows over 256 columns (fp8 decode + fp32 partial sums + the three-rounding rescale on the host)
    r0 = 97 * 32
    rows = slice(r0, r0 + 32)
    srows = slice(r0 // qm, r0 // qm + 32 // qm)
    nb = 256 // (S // sB.shape[1])
    wo = oracle.scaled_mm(A[rows].cpu().contiguous(), B[:, :256].cpu(), sA[srows].cpu().contiguous(), sB[:, :nb].cpu().contiguous())
    assert_fp8_close(out[rows][:, :256], wo, mag[rows][:, :256].cpu())


@pytest.mark.parametrize("S", [1024, 2048])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_auto_dispatch_at_benchmark_mid_sizes(S, layout):
    """1024^3 and 2048^3 of benchmark_mm.py go through plan_gemm's OTHER choices (128 x 128 tiles, split-K): int8 bit-exact against the
    oracle on sampled rows and against the unfused epilogue everywhere; fp8 within the stated tolerance everywhere."""
    torch.manual_seed(2025 * 2024 + S)
    acm, bcm = LAYOUTS[layout]
    A = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), acm)
    B = to_layout(torch.randint(-128, 127, (S, S), dtype=torch.int8, device=DEV), bcm)
    rs = torch.randn(S, 1, device=DEV).div(128).bfloat16()
    cs = torch.randn(1, S, device=DEV).div(128).bfloat16()
    C = gn_kernels.int8_mm(A, B)
    rows = torch.randperm(S)[:32].sort().values.to(DEV)
    assert torch.equal(C[rows].cpu(), oracle.int8_mm(A[rows].cpu().contiguous(), B.cpu()))
    assert torch.equal(C.sum(0, dtype=torch.int64), (A.sum(0, dtype=torch.int64).double() @ B.double()).to(torch.int64))
    out = gn_kernels.scaled_mm(A, B, rs, cs)
    assert bits_equal(out, ((C.float() * rs.float()) * cs.float()).bfloat16())
    assert bits_equal(out[rows], oracle.scaled_mm(A[rows].cpu().contiguous(), B.cpu(), rs[rows].cpu().contiguous(), cs.cpu()))
    tsA = to_layout(torch.randn(S, S // 128, device=DEV).div(128).bfloat16(), acm)
    tsB = to_layout(torch.randn(S // 128, S // 128, device=DEV).div(128).bfloat16(), bcm)
    want_t = oracle.scaled_mm(A[rows].cpu().contiguous(), B.cpu(), tsA[rows].cpu().contiguous(), tsB.cpu())
    with tile_rescale("exact"):
        assert bits_equal(gn_kernels.scaled_mm(A, B, tsA, tsB)[rows], want_t)
    torch.testing.assert_close(gn_kernels.scaled_mm(A, B, tsA, tsB)[rows].cpu(), want_t, atol=1e-4, rtol=1e-2)
    A8 = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), acm)
    B8 = to_layout(torch.randn(S, S, device=DEV).to(torch.float8_e4m3fn), bcm)
    out8 = gn_kernels.scaled_mm(A8, B8, rs, cs)
    want8 = _scaled_ref(A8, B8, rs, cs)
    mag8 = (A8.float().abs() * rs.float().abs()) @ (B8.float().abs() * cs.float().abs())
    assert not bool(((out8.float() - want8).abs() > 1.6e-2 * want8.abs() + 5e-5 * mag8 + 1e-5).any())
    failing = _unwidened_failures(out8, want8)
    _record(f"scaled_fp8_{S}_{layout}", {"failing": failing, "of": S * S})
    assert failing == 0, "scaled fp8 must pass the reference's own unwidened assert_close (benchmark_mm.py:142-143)"


def test_tile_scaled_refuses_ragged_scale_tilings():
    """M = 100 with 3 scale rows gives QM = 33 and row 99 would index scale row 3 (out of bounds, in the reference too)."""
    A = torch.zeros(100, 128, dtype=torch.int8, device=DEV)
    B = torch.zeros(128, 128, dtype=torch.int8, device=DEV)
    with pytest.raises(RuntimeError, match="evenly"):
        gn_kernels.scaled_mm(A, B, torch.ones(3, 1, device=DEV), torch.ones(1, 2, device=DEV))
    ok = gn_kernels.scaled_mm(A, B, torch.ones(4, 1, device=DEV), torch.ones(1, 2, device=DEV))
    assert ok.shape == (100, 128)


# ------------------------------------------------------------------------------- K tails on the MFMA kernel
def _padded(x, column_major, pad_to=16):
    """x on the device in the given layout inside a buffer whose leading dimension is a multiple of 16 bytes (what the MFMA path needs);
    the padding holds 0x7f garbage (a NaN in e4m3) that must never reach a result."""
    src = x.T if column_major else x
    rows, cols = src.shape
    ld = (cols + pad_to - 1) // pad_to * pad_to + pad_to
    buf = torch.full((rows + 1, ld), 0x7f, dtype=torch.uint8, device=DEV).view(x.dtype)
    buf[:rows, :cols] = src.to(DEV)
    v = buf[:rows, :cols]
    return v.T if column_major else v


@pytest.mark.parametrize("K", [1, 16, 48, 100, 127, 130, 200, 333, 1000, 16448])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_k_tails_run_on_the_mfma_kernel_bit_exact(K, layout):
    """The reference takes any even K (triton_mm.py:140,404; SURVEY 7.2 names ViT's K = 16448): the 128 x 128 MFMA kernel masks the last,
    partial K-step in LDS.  Forced onto that kernel (the call fails if the shape were not eligible), int8 bit-exact against the oracle,
    fp8 within the stated tolerance; the garbage beside the operands (row padding, rows beyond K) must not leak."""
    M, N = (144, 112) if K < 10000 else (128, 128)
    g = torch.Generator().manual_seed(K)
    acm, bcm = LAYOUTS[layout]
    A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g)
    B = torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g)
    rs = torch.randn(M, 1, generator=g).div(128).bfloat16()
    cs = torch.randn(1, N, generator=g).div(128).bfloat16()
    Ad, Bd = _padded(A, acm), _padded(B, bcm)
    L = _cabi.load()
    L.gnk_set_gemm_variant(VARIANTS["tile128"])
    try:
        assert torch.equal(gn_kernels.int8_mm(Ad, Bd).cpu(), oracle.int8_mm(A, B))
        assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, rs.to(DEV), cs.to(DEV)), oracle.scaled_mm(A, B, rs, cs))
        if K <= 1000:
            A8 = torch.randn(M, K, generator=g).to(torch.float8_e4m3fn)
            B8 = torch.randn(K, N, generator=g).to(torch.float8_e5m2)
            got = gn_kernels.scaled_mm(_padded(A8, acm), _padded(B8, bcm), rs.to(DEV), cs.to(DEV))
            assert_fp8_close(got, oracle.scaled_mm(A8, B8, rs, cs), fp8_mag(A8, B8, rs, cs))
    finally:
        L.gnk_set_gemm_variant(-1)
    # the heuristic must take an MFMA kernel too (same bits), not the generic fallback: compare, and for the big K time-box it
    out = gn_kernels.scaled_mm(Ad, Bd, rs.to(DEV), cs.to(DEV))
    assert bits_equal(out, oracle.scaled_mm(A, B, rs, cs))


@pytest.mark.parametrize("shape", [(256, 512, 1024), (1000, 330, 2048), (64, 48, 4100), (2048, 2048, 8192), (1, 7, 64)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_scaled_mm_t_is_the_materialised_transpose(shape, dt):
    """gn_kernels.scaled_mm_t (the grad-weight form of subclasses/mixed_precision.py): bit-identical to scaled_mm(...).T.contiguous() --
    the operand swap comes with the epilogue's multiply order swapped, so each element is still (acc * scale_A[m]) * scale_B[n] -- in the
    grad-weight layout (A = X.T, both operands M/N-major), with K-major operands, with ragged sizes, a K tail and a split-K shape."""
    import gn_kernels

    M, N, K = shape
    torch.manual_seed(M + N + K)
    sa = (torch.rand(M, 1, device=DEV) * 0.02 + 1e-3).to(dt)
    sb = (torch.rand(1, N, device=DEV) * 0.02 + 1e-3).to(dt)
    At = torch.randint(-128, 128, (K, M), dtype=torch.int8, device=DEV)          # A = At.T: M-major, as X.T in the Linear backward
    B = torch.randint(-128, 128, (K, N), dtype=torch.int8, device=DEV)
    for A in (At.T, At.T.contiguous()):
        want = gn_kernels.scaled_mm(A, B, sa, sb).T.contiguous()
        got = gn_kernels.scaled_mm_t(A, B, sa, sb)
        assert got.shape == (N, M) and got.is_contiguous() and got.dtype is dt
        assert torch.equal(got.view(torch.int16 if dt is torch.bfloat16 else torch.int32), want.view(torch.int16 if dt is torch.bfloat16 else torch.int32))
    a8 = torch.randn(M, K, device=DEV).to(torch.float8_e4m3fn)
    b8 = torch.randn(N, K, device=DEV).to(torch.float8_e4m3fn).T
    want = gn_kernels.scaled_mm(a8, b8, sa, sb).T.contiguous()
    got = gn_kernels.scaled_mm_t(a8, b8, sa, sb)
    torch.testing.assert_close(got.float(), want.float(), rtol=1.6e-2, atol=1e-5 + 5e-5 * float((a8.float().abs() @ b8.float().abs()).max() * sa.float().max() * sb.float().max()))


@pytest.mark.parametrize("N,slabs", [(2048, 4), (5632, 2)])
@pytest.mark.parametrize("layout", ["A@B.T", "A.T@B"])
def test_split_k_on_the_256_kernel_is_bit_identical(layout, N, slabs):
    """2048 x 2048 x 32768 (the grad-weight GEMM of a square Llama-1B Linear: 64 tiles of 256 x 256 on 256 CUs): the planner cuts K into four
    chunks on the four-stage 256 x 256 kernel (int32 slabs + reduction kernel).  Integer partial sums add exactly: the result equals the
    unsplit 128 x 128 kernel's bit for bit, in the forward layout and in the grad-weight layout (both operands M/N-major)."""
    import gn_kernels
    from gn_kernels import _cabi

    L = _cabi.load()
    # N = 5632 (the MLP grad-weights: 176 tiles): an UNEVEN two-way split -- the long first splits hold 176 CUs, the short second ones
    # run in rounds on the other 80
    M = 2048
    K = 32768
    assert L.gnk_scaled_mm_workspace_bytes(_cabi.I8, _cabi.I8, M, N, K) == slabs * M * N * 4
    torch.manual_seed(1)
    if layout == "A@B.T":
        A = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=DEV)
        B = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV).T
    else:
        A = torch.randint(-128, 128, (K, M), dtype=torch.int8, device=DEV).T
        B = torch.randint(-128, 128, (K, N), dtype=torch.int8, device=DEV)
    sa = (torch.rand(M, 1, device=DEV) * 1e-3 + 1e-4).bfloat16()
    sb = (torch.rand(1, N, device=DEV) * 1e-3 + 1e-4).bfloat16()
    got = gn_kernels.scaled_mm(A, B, sa, sb)
    got_t = gn_kernels.scaled_mm_t(A, B, sa, sb)
    old = L.gnk_set_gemm_variant(1)
    try:
        want = gn_kernels.scaled_mm(A, B, sa, sb)
    finally:
        L.gnk_set_gemm_variant(old)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(got_t.view(torch.int16), want.T.contiguous().view(torch.int16))


def test_single_row_operand_with_unpadded_k_tail_stays_inside_its_buffer():
    """an operand with one row (M == 1 or N == 1) has no pitch to round K up to, so with K % 16 != 0 the K-tail stager's 16-byte
    chunk would read past the end of an exact-size buffer.  Such calls take the generic kernel.  The operands sit at the very end of their
    allocations (a raw C-ABI caller could have a page edge there); results against the oracle."""
    g = torch.Generator().manual_seed(7)
    for M, N, K in ((1, 7, 100), (5, 1, 100), (1, 1, 33)):
        A = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g)
        B = torch.randint(-128, 128, (K, N), dtype=torch.int8, generator=g)
        rs, cs = torch.rand(M, 1, generator=g) + 0.5, torch.rand(1, N, generator=g) + 0.5
        bufA = torch.empty(4096 + M * K, dtype=torch.int8, device=DEV)
        Ad = bufA[4096:].view(M, K).copy_(A)                       # ends exactly at the end of the allocation's requested size
        bufB = torch.empty(4096 + K * N, dtype=torch.int8, device=DEV)
        Bd = bufB[4096:].view(N, K).copy_(B.T).T                    # column-major (K contiguous), same placement
        assert torch.equal(gn_kernels.int8_mm(Ad, Bd).cpu(), oracle.int8_mm(A, B.T.contiguous().T))
        assert bits_equal(gn_kernels.scaled_mm(Ad, Bd, rs.to(DEV), cs.to(DEV)), oracle.scaled_mm(A, B.T.contiguous().T, rs, cs))
