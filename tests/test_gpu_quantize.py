"""GPU parity of the INT8 quantize kernels (csrc/gnk_quantize.hip) through the gn_kernels op -> C ABI.
Bit-exact against (1) goldens generated from the reference, (2) the CPU oracle on seeded inputs,
(3) size-independent properties at the benchmark's full size."""
import pytest
import torch

import gn_kernels
import oracle
from gn_kernels import quantize as qmod
from subclasses import quantize_int8
from tests._util import FLOAT_DTYPES, bits_equal, load, t_from_bits

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _check_vs_oracle(x_cpu, dim, sr=False, eps=1e-12):
    xg = x_cpu.to(DEV) if x_cpu.is_contiguous() else x_cpu.T.contiguous().to(DEV).T
    assert xg.stride() == x_cpu.stride()
    if sr:
        # the op's Philox key is the next draw of torch's CPU generator (gn_kernels/quantize.py:draw_rng): peek it, then rewind
        state = torch.get_rng_state()
        seed = int(qmod.draw_rng(True).item())
        torch.set_rng_state(state)
        qo, so = oracle.quantize_int8(x_cpu, True, dim=dim, eps=eps, seed=seed, offset=0)
    else:
        qo, so = oracle.quantize_int8(x_cpu, dim=dim, eps=eps)
    q, s = quantize_int8(xg, sr, dim=dim, eps=eps)
    assert q.dtype is torch.int8 and q.stride() == xg.stride() and s.dtype is x_cpu.dtype and s.shape == so.shape
    assert torch.equal(q.cpu(), qo), f"int8 mismatch: {(q.cpu() != qo).sum().item()} elements"
    assert bits_equal(s, so)


@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
def test_golden_from_reference(name):
    g = load("g1_quantize_int8.npz")
    dt = FLOAT_DTYPES[name]
    x = t_from_bits(g[f"x_{name}"], dt).to(DEV)
    q, s = quantize_int8(x, dim=1)
    assert torch.equal(q.cpu(), torch.from_numpy(g[f"q_{name}"]))
    assert bits_equal(s, t_from_bits(g[f"s_{name}"], dt).view(64, 1))
    q0, s0 = quantize_int8(x, dim=0)                       # strided reduction, layout kept
    assert q0.stride() == x.stride()
    assert torch.equal(q0.cpu(), torch.from_numpy(g[f"q_dim0_{name}"]))
    assert bits_equal(s0, t_from_bits(g[f"s_dim0_{name}"], dt).view(1, 256))
    qt, st = quantize_int8(x.T, dim=1)                     # the B.T spelling of mixed_precision.py:102
    assert qt.stride() == x.T.stride() and torch.equal(qt.T, q0) and bits_equal(st.view(-1), s0.view(-1))
    qe, se = quantize_int8(x, dim=1, eps=1e-5) if name == "bf16" else (None, None)
    if qe is not None:
        assert torch.equal(qe.cpu(), torch.from_numpy(g["q_eps1e5_bf16"]))


@pytest.mark.parametrize("shape", [(37, 8), (64, 256), (33, 1024), (16, 4096), (8, 8192), (3, 16384 + 512), (5, 40000)])
@pytest.mark.parametrize("name", ["bf16", "f32"])
def test_rowwise_vs_oracle(shape, name):
    torch.manual_seed(shape[1])
    x = (torch.randn(shape) * torch.rand(shape[0], 1) * 4).to(FLOAT_DTYPES[name])
    x[0] = 0
    _check_vs_oracle(x, 1)


@pytest.mark.parametrize("shape", [(128, 64), (200, 256), (1000, 1024), (77, 4096)])
@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
def test_colwise_vs_oracle(shape, name):
    torch.manual_seed(shape[0])
    x = (torch.randn(shape) * torch.rand(1, shape[1]) * 4).to(FLOAT_DTYPES[name])
    x[:, 3] = 0
    _check_vs_oracle(x, 0)            # reduce over rows of a row-major matrix
    _check_vs_oracle(x.T, 1)          # same thing spelled as a transposed view


@pytest.mark.parametrize("shape,dim", [((7, 13), 1), ((7, 13), 0), ((1, 1), 1), ((130, 5), 0), ((5, 130), 1), ((64, 100), 0)])
def test_generic_fallback_odd_shapes_vs_oracle(shape, dim):
    torch.manual_seed(7)
    _check_vs_oracle(torch.randn(shape).bfloat16(), dim)


def test_non_dense_strides_and_nd_inputs():
    torch.manual_seed(8)
    big = torch.randn(32, 300).bfloat16()
    xs = big[:, 10:138]                                  # unaligned start, row pitch 300
    q, s = quantize_int8(xs.to(DEV)[:, :], dim=1)
    qo, so = oracle.quantize_int8(xs.contiguous(), dim=1)
    assert torch.equal(q.cpu(), qo) and bits_equal(s, so)
    sl = big.to(DEV)[:, 10:138]
    q2, s2 = quantize_int8(sl, dim=1)
    assert torch.equal(q2.cpu(), qo) and bits_equal(s2, so)
    x3 = torch.randn(4, 6, 64).bfloat16()
    q3, s3 = quantize_int8(x3.to(DEV))
    qo3, so3 = oracle.quantize_int8(x3.view(24, 64), dim=1)
    assert q3.shape == (4, 6, 64) and s3.shape == (4, 6, 1)
    assert torch.equal(q3.cpu().view(24, 64), qo3) and bits_equal(s3.view(24, 1), so3)
    qm, sm = quantize_int8(x3.to(DEV), dim=1)            # inner dim: moved last, quantized, moved back
    assert qm.shape == (4, 6, 64) and sm.shape == (4, 1, 64)
    e = torch.empty(0, 16, dtype=torch.bfloat16, device=DEV)
    qe, se = quantize_int8(e)
    assert qe.shape == (0, 16) and se.shape == (0, 1)


@pytest.mark.parametrize("shape,dim", [((64, 512), 1), ((256, 128), 0), ((9, 33), 1),
                                       ((130, 256), 0),      # column pass whose 4-row groups do not start a Philox block (P % 4 != 0)
                                       ((1000, 512), 0),     # column pass with a ragged last row slab
                                       ((3, 16384), 1),      # rows too long for the register-resident kernel (two-pass kernel)
                                       ((33, 8192), 1)])
def test_stochastic_rounding_bit_exact_vs_oracle_philox(shape, dim):
    """Four elements share a Philox block (word e & 3 of block e >> 2): the kernels draw them together where a thread holds
    four consecutive logical elements and fall back to per-element draws elsewhere -- same stream either way."""
    torch.manual_seed(1234)
    for dt in (torch.bfloat16, torch.float32):
        _check_vs_oracle(torch.randn(shape).to(dt), dim, sr=True)
    _check_vs_oracle(torch.randn(shape).bfloat16(), dim, sr=True)      # a later call: another key, still the oracle's stream


def test_stochastic_rounding_is_unbiased_like_the_reference():
    """Parity with the reference's torch.rand_like SR is statistical: E[q * scale] == x."""
    torch.manual_seed(0)
    x = torch.full((8192, 8), 0.3, device=DEV)
    x[:, 0] = 127.0                                      # scale exactly 1
    acc = torch.zeros(8192, 8, device=DEV)
    for _ in range(8):
        q, _ = quantize_int8(x, True)
        acc += q.float()
    mean = (acc / 8)[:, 1:].mean().item()
    assert abs(mean - 0.3) < 5e-3, mean
    q1, _ = quantize_int8(x, True)
    q2, _ = quantize_int8(x, True)
    assert not torch.equal(q1, q2)                       # successive calls draw fresh noise


@pytest.mark.parametrize("dim", [1, 0])
def test_full_size_properties_4096(dim):
    """Benchmark size (4096 x 4096 bf16): scale == amax/127, |x - q*scale| <= scale/2, fixed point."""
    torch.manual_seed(2025 * 2024)
    x = torch.randn(4096, 4096, device=DEV).bfloat16()
    q, s = quantize_int8(x, dim=dim)
    xf = x.float()
    s32 = xf.abs().amax(dim, keepdim=True) / 127
    assert bits_equal(s, s32.to(torch.bfloat16))
    assert int(q.min()) >= -127 and int(q.max()) <= 127
    assert int(q.abs().amax(dim).min()) == 127           # every row / column hits +-127
    err = (xf - q.float() * s32).abs()
    assert bool((err <= s32 * 0.50001).all())            # 0.5 step + fp32 division rounding at |x/s| <= 127
    # full-size bit-exact check against the ATen restatement run on the HOST (IEEE division); the same
    # formula run by ATen on the device is not a bit-exact witness on ROCm (its fp32 division differs
    # from the correctly rounded one in rare cases), so it only has to agree almost everywhere
    from oracle import torch_ref

    want_q, want_s = torch_ref.quantize_int8(x.cpu(), dim=dim)
    assert torch.equal(q.cpu(), want_q) and bits_equal(s, want_s)
    dev_formula = torch.round(xf / s32.clamp_min(1e-12)).clamp(-128, 127).to(torch.int8)
    assert (q != dev_formula).float().mean().item() < 1e-4
    q2, _ = quantize_int8((q.float() * s32), dim=dim)    # quantizing the dequantized tensor is a fixed point
    assert torch.equal(q2, q)


# ------------------------------------------------------------------------------- fused dual-direction pair (8f rank 1)
@pytest.mark.parametrize("shape", [(8, 8), (64, 256), (37, 512), (1000, 1024), (4099, 2048), (515, 4096), (300, 5632), (129, 6144), (64, 3072)])
@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
def test_rowwise_colmax_and_colwise_given_equal_the_separate_passes(shape, name):
    """quantize_int8_rowwise_colmax == quantize_int8(dim=1) plus the exact fp32 column absmax; feeding that absmax to
    quantize_int8_colwise_given == quantize_int8(dim=0): bit for bit, and both equal the CPU oracle."""
    dt = FLOAT_DTYPES[name]
    if name == "f32" and shape[1] > 3072:
        pytest.skip("fused kernel holds rows of at most 3072 fp32 elements")
    torch.manual_seed(shape[0] * 31 + shape[1])
    x = (torch.randn(shape) * torch.rand(shape[0], 1) * 4 * torch.rand(1, shape[1])).to(dt)
    x[0] = 0
    x[:, 1] = 0                                # an all-zero column: absmax 0, scale 0, data 0
    x[3, 2] = -7.5                             # column whose maximum is a negative value
    xg = x.to(DEV)
    assert qmod.supports_rowwise_colmax(xg) and qmod.supports_colwise_given(xg)
    q, s, cm = qmod.quantize_int8_rowwise_colmax(xg)
    q1, s1 = quantize_int8(xg, dim=1)
    assert torch.equal(q, q1) and bits_equal(s, s1)
    assert cm.dtype is torch.float32 and torch.equal(cm.cpu(), x.float().abs().amax(0))
    q0, s0 = qmod.quantize_int8_colwise_given(xg, cm)
    q0r, s0r = quantize_int8(xg, dim=0)
    assert torch.equal(q0, q0r) and bits_equal(s0, s0r) and q0.stride() == xg.stride()
    qo, so = oracle.quantize_int8(x, dim=1)
    assert torch.equal(q.cpu(), qo) and bits_equal(s, so)
    qo0, so0 = oracle.quantize_int8(x, dim=0)
    assert torch.equal(q0.cpu(), qo0) and bits_equal(s0, so0)


def test_rowwise_colmax_padded_rows_and_fallback_shapes():
    """Row-padded views (stride > C) take the fused kernels; rows longer than the kernel holds report unsupported."""
    torch.manual_seed(5)
    base = torch.randn(300, 1024 + 64, device=DEV).bfloat16()
    xg = base[:, :1024]
    assert qmod.supports_rowwise_colmax(xg)
    q, s, cm = qmod.quantize_int8_rowwise_colmax(xg)
    q1, s1 = quantize_int8(xg.contiguous(), dim=1)
    assert torch.equal(q, q1) and bits_equal(s, s1) and torch.equal(cm, xg.float().abs().amax(0))
    q0, s0 = qmod.quantize_int8_colwise_given(xg, cm)
    q0r, s0r = quantize_int8(xg.contiguous(), dim=0)
    assert torch.equal(q0, q0r) and bits_equal(s0, s0r)
    assert not qmod.supports_rowwise_colmax(torch.zeros(4, 8192, device=DEV).bfloat16())
    assert not qmod.supports_rowwise_colmax(torch.zeros(8, 16, device=DEV).bfloat16().T)
    with pytest.raises(RuntimeError):
        qmod.quantize_int8_rowwise_colmax(torch.zeros(4, 8192, device=DEV).bfloat16())


def test_rowwise_colmax_stochastic_rounding_statistics():
    """SR through the fused kernels: unbiased (mean of dequantized == x within the CLT bound), both directions."""
    torch.manual_seed(11)
    x = torch.randn(256, 512, device=DEV).bfloat16()
    acc_r = torch.zeros_like(x, dtype=torch.float32)
    acc_c = torch.zeros_like(x, dtype=torch.float32)
    n = 64
    for _ in range(n):
        q, s, cm = qmod.quantize_int8_rowwise_colmax(x, True)
        acc_r += q.float() * s.float()
        qc, sc = qmod.quantize_int8_colwise_given(x, cm, True)
        acc_c += qc.float() * sc.float()
    for acc, sc_ in ((acc_r, s), (acc_c, sc)):
        err = (acc / n - x.float()).abs() / sc_.float().clamp_min(1e-12)      # in units of one quantization step
        assert err.max().item() < 0.5 and err.mean().item() < 0.08           # sd of the mean <= 0.5 / sqrt(64) = 0.0625


def test_full_size_fused_pair_32768x2048():
    """Llama-1B activation shape: fused pair == separate passes, and the column absmax is exact."""
    torch.manual_seed(2025)
    x = torch.randn(32768, 2048, device=DEV).bfloat16()
    q, s, cm = qmod.quantize_int8_rowwise_colmax(x)
    q1, s1 = quantize_int8(x, dim=1)
    assert torch.equal(q, q1) and bits_equal(s, s1) and torch.equal(cm, x.float().abs().amax(0))
    q0, s0 = qmod.quantize_int8_colwise_given(x, cm)
    q0r, s0r = quantize_int8(x, dim=0)
    assert torch.equal(q0, q0r) and bits_equal(s0, s0r)


@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
def test_nonfinite_inputs_surface_like_the_reference(name):
    """fmaxf-based absmax dropped NaN and quantized NaN rows with a finite scale.  g9 (reference on the CPU): NaN -> NaN scale,
    Inf -> Inf scale and zeros; every kernel family (row-wise, strided two-pass, fused pair, long rows, generic) against it."""
    g = load("g9_quantize_nonfinite.npz")
    dt = FLOAT_DTYPES[name]
    x = t_from_bits(g[f"x_{name}"], dt).to(DEV)

    def check(q, s, dim):
        want_s = t_from_bits(g[f"s_dim{dim}_{name}"], dt).float()
        got = s.float().cpu().view(want_s.shape)
        assert torch.equal(torch.isnan(got), torch.isnan(want_s))
        assert torch.equal(got[~torch.isnan(want_s)], want_s[~torch.isnan(want_s)])
        undef = torch.from_numpy(g[f"undef_dim{dim}_{name}"])
        qc = q.cpu()
        assert torch.equal(qc[~undef], torch.from_numpy(g[f"q_dim{dim}_{name}"])[~undef])
        assert int(qc[undef].abs().max()) == 0

    check(*quantize_int8(x, dim=1), 1)                       # row-wise register-resident kernel
    check(*quantize_int8(x, dim=0), 0)                       # strided reduction (absmax pass + quantize pass)
    q, s, cm = qmod.quantize_int8_rowwise_colmax(x)          # fused pair
    check(q, s, 1)
    check(*qmod.quantize_int8_colwise_given(x, cm), 0)
    xo = torch.zeros(24, 520, device=DEV, dtype=dt)[:, 3:515]   # unaligned base: generic kernel
    xo.copy_(x)
    check(*quantize_int8(xo, dim=1), 1)
    check(*quantize_int8(xo, dim=0), 0)
    # long rows (two-pass row kernel): tile the 512 columns to 16896 > 16384
    xl = x.repeat(1, 33)
    ql, sl = quantize_int8(xl, dim=1)
    check(ql[:, :512], sl, 1)


@pytest.mark.parametrize("shape", [(5, 64), (130, 256), (1000, 1024), (2049, 512), (8192, 1024), (4100, 72)])
@pytest.mark.parametrize("name", ["bf16", "f16", "f32"])
def test_strided_reduction_both_workspace_forms_vs_oracle(shape, name):
    """gnk_quantize_int8 along the strided dim runs pass 1 either as per-block partial maxima (the workspace its _workspace_bytes asks
    for: nothing to clear, no atomics) or, given only the minimum of one word per logical row, as atomicMax into cleared words.  Both
    through the C ABI, both bit-equal to the oracle; the partial form on a workspace full of garbage."""
    from gn_kernels import _cabi

    L = _cabi.load()
    dt = FLOAT_DTYPES[name]
    torch.manual_seed(shape[0] * 7 + shape[1])
    x_cpu = (torch.randn(shape) * torch.rand(1, shape[1]) * 8).to(dt)
    x_cpu[:, 3] = 0
    x_cpu[shape[0] - 1, 7] = 300.0                              # a column whose maximum sits in the last (ragged) row block
    qo, so = oracle.quantize_int8(x_cpu, dim=0)
    x = x_cpu.to(DEV)
    P, Q = shape                                                # logical rows = the Q columns, reduced over P
    full = L.gnk_quantize_int8_workspace_bytes(Q, P, 1, x.stride(0))
    assert full >= Q * 4 and full % (Q * 4) == 0 and full // (Q * 4) <= 32
    for nbytes in (full, Q * 4):
        ws = torch.full((full,), 0xFF, dtype=torch.uint8, device=DEV)        # garbage (NaN patterns): a form that trusts it would fail
        q = torch.empty_like(x, dtype=torch.int8)
        s = torch.empty(Q, dtype=dt, device=DEV)
        rc = L.gnk_quantize_int8(x.data_ptr(), _cabi.DTYPE_CODE[dt], Q, P, 1, x.stride(0), q.data_ptr(), 1, q.stride(0), s.data_ptr(), 1e-12,
                                 0, 0, 0, ws.data_ptr(), nbytes, _cabi.stream_of(x))
        assert rc == 0, L.gnk_last_error()
        torch.cuda.synchronize()
        assert torch.equal(q.cpu(), qo), f"int8 mismatch with a {nbytes}-byte workspace"
        assert bits_equal(s.view(1, -1), so)
    rc = L.gnk_quantize_int8(x.data_ptr(), _cabi.DTYPE_CODE[dt], Q, P, 1, x.stride(0), q.data_ptr(), 1, q.stride(0), s.data_ptr(), 1e-12,
                             0, 0, 0, ws.data_ptr(), Q * 4 - 1, _cabi.stream_of(x))
    assert rc != 0 and b"workspace" in L.gnk_last_error()
