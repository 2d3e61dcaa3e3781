"""GPU parity of the BitNet path (csrc/gnk_bitnet.hip) through the gn_kernels ops -> C ABI and through subclasses/bitnet.py.

Bars: ternary quantize, 2-bit pack and unpack BIT-EXACT against the goldens generated from the reference
(tests/golden/g8_bitnet.npz) and against the numpy oracle on seeded inputs; the abs-mean scale to 1e-6 relative (fp32 sum,
order is the implementation's); Linear forward bit-exact against the reference golden (integer GEMM + scale chain);
backward products are bf16 library matmuls: reference's own bf16 tolerance (rtol 1.6e-2, atol 2e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from gn_kernels import _cabi
from gn_kernels import bitnet as bop
from oracle import bitnet_ref
from subclasses import bitnet as bn
from tests._util import bits_equal, load, t_from_bits

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
WEIGHTS = {"bf16_40x72": BF, "bf16_5x12": BF, "f32_7x20": torch.float32, "bf16_256x384": BF}


@pytest.mark.parametrize("name", list(WEIGHTS))
def test_golden_scale_quantize_pack(name):
    g = load("g8_bitnet.npz")
    w = t_from_bits(g[f"w_{name}"], WEIGHTS[name]).to(DEV)
    want_s = float(g[f"s_{name}"])
    s = bop.bitnet_absmean(w)
    assert s.dtype is torch.float32 and s.shape == () and abs(s.item() - want_s) <= 1e-6 * want_s
    gs = torch.tensor(want_s, device=DEV)
    assert np.array_equal(bop.bitnet_quantize_weight(w, gs).cpu().numpy(), g[f"q_{name}"])
    assert np.array_equal(bop.bitnet_quantize_weight(w, gs, packed=True).cpu().numpy(), g[f"p_{name}"])
    assert np.array_equal(bop.bitnet_pack_i2(torch.from_numpy(g[f"q_{name}"]).to(DEV)).cpu().numpy(), g[f"p_{name}"])
    assert np.array_equal(bop.bitnet_unpack_i2(torch.from_numpy(g[f"p_{name}"]).to(DEV)).cpu().numpy(), g[f"q_{name}"])


def test_golden_ties_floor_and_arbitrary_bytes():
    g = load("g8_bitnet.npz")
    for tag, dt in (("ties", BF), ("tiny", torch.float32)):
        w = t_from_bits(g[f"w_{tag}"], dt).to(DEV)
        s = torch.tensor(float(g[f"s_{tag}"]), device=DEV)
        assert np.array_equal(bop.bitnet_quantize_weight(w, s).cpu().numpy(), g[f"q_{tag}"])
    assert np.array_equal(bop.bitnet_pack_i2(torch.from_numpy(g["pack_in"]).to(DEV)).cpu().numpy(), g["pack_out"])
    assert np.array_equal(bop.bitnet_unpack_i2(torch.from_numpy(g["unpack_in"]).to(DEV)).cpu().numpy(), g["unpack_out"])


@pytest.mark.parametrize("shape", [(1, 4), (3, 20), (129, 1028), (1024, 1024), (2048, 5632), (4096, 11008)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_vs_oracle_seeded(shape, dt):
    torch.manual_seed(shape[0] * 7 + shape[1])
    w = (torch.randn(shape) * 0.02).to(dt)
    w[0, :2] = 0
    wd = w.to(DEV)
    s = bop.bitnet_absmean(wd)
    so = bitnet_ref.get_bitnet_scale(w.float().numpy())
    assert abs(s.item() - float(so)) <= 1e-6 * float(so)
    assert bits_equal(bop.bitnet_absmean(wd), s)                              # fixed summation order: same bits every call
    q = bop.bitnet_quantize_weight(wd, s)
    qo = bitnet_ref.quantize_bitnet_weight(w.float().numpy(), np.float32(s.item()))
    assert np.array_equal(q.cpu().numpy(), qo)
    p = bop.bitnet_quantize_weight(wd, s, packed=True)
    assert p.shape == (shape[0], shape[1] // 4) and np.array_equal(p.cpu().numpy(), bitnet_ref.pack_i2(qo))
    assert torch.equal(bop.bitnet_pack_i2(q), p)                              # fused form == quantize then pack
    assert torch.equal(bop.bitnet_unpack_i2(p), q)                            # encode -> decode round trip at full size
    frac = (q != 0).float().mean().item()
    assert 0.4 < frac < 0.8 or shape[0] * shape[1] < 100                      # abs-mean scale keeps roughly 2/3 of a Gaussian non-zero


def test_big_scale_small_scale_and_bad_inputs():
    w = torch.randn(64, 64, device=DEV)
    for s in (1e-9, 1e30, float("inf")):
        st = torch.tensor(s, device=DEV)
        want = bitnet_ref.quantize_bitnet_weight(w.cpu().numpy(), np.float32(s))
        assert np.array_equal(bop.bitnet_quantize_weight(w, st).cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        bop.bitnet_absmean(w.T)                                               # not contiguous: the op does not copy
    with pytest.raises(RuntimeError):
        bop.bitnet_quantize_weight(w, torch.tensor(1.0))                      # host scale
    with pytest.raises(AssertionError):
        bop.bitnet_quantize_weight(w[:, :62].contiguous(), torch.tensor(1.0, device=DEV), packed=True)
    assert bn.get_bitnet_scale(w.T).item() == pytest.approx(bop.bitnet_absmean(w).item(), rel=1e-6)   # the module copies for the caller
    L = _cabi.load()
    assert L.gnk_bitnet_pack_i2(w.data_ptr(), 6, w.data_ptr(), None) < 0 and b"multiple of 4" in L.gnk_last_error()
    assert L.gnk_bitnet_absmean(w.data_ptr(), 0, 16, w.data_ptr(), w.data_ptr(), 8, None) == -4          # GNK_ERR_WORKSPACE


def _close(got, want):
    torch.testing.assert_close(got.float().cpu(), want.float(), rtol=1.6e-2, atol=2e-3)


def test_training_linear_matches_the_reference_golden():
    g = load("g8_bitnet.npz")
    lin = nn.Linear(256, 128, bias=True).to(BF)
    with torch.no_grad():
        lin.weight.copy_(t_from_bits(g["train_W"], BF))
        lin.bias.copy_(t_from_bits(g["train_b"], BF))
    lin = bn.convert_bitnet(lin.to(DEV))
    assert isinstance(lin.weight, bn.BitNetTrainingLinearWeight)
    x = t_from_bits(g["train_x"], BF).to(DEV).requires_grad_()
    y = lin(x)
    y.backward(t_from_bits(g["train_gy"], BF).to(DEV))
    assert bits_equal(y, t_from_bits(g["train_y"], BF))
    _close(x.grad, t_from_bits(g["train_dx"], BF))
    gw = lin.weight.grad
    _close(gw._data if isinstance(gw, bn.BitNetTrainingLinearWeight) else gw, t_from_bits(g["train_dW"], BF))
    _close(lin.bias.grad, t_from_bits(g["train_db"], BF))


def test_packed_2bit_linear_matches_the_reference_golden():
    g = load("g8_bitnet.npz")
    w = bn.BitNetPacked2bitLinearWeight(torch.from_numpy(g["packed_i2"]).to(DEV), t_from_bits(g["packed_scale"], BF).reshape(()).to(DEV))
    assert bits_equal(w.dequantize(), t_from_bits(g["packed_deq"], BF))
    x = t_from_bits(g["packed_x"], BF).to(DEV).requires_grad_()
    y = F.linear(x, w)
    y.backward(t_from_bits(g["packed_gy"], BF).to(DEV))
    assert bits_equal(y, t_from_bits(g["packed_y"], BF))
    _close(x.grad, t_from_bits(g["packed_dx"], BF))


def test_fsdp_hook_pair_on_device_and_llama_sized_step():
    """pre-all-gather emits the fused packed codes; post-all-gather's container reproduces the training weight's forward."""
    torch.manual_seed(0)
    lin = bn.convert_bitnet(nn.Linear(2048, 5632, bias=False).to(BF).to(DEV))
    w = lin.weight
    scale = bn.get_bitnet_scale(w._data)
    codes = bn._quantize_and_pack(w._data, scale)
    assert codes.shape == (5632, 512) and codes.dtype is torch.int8
    packed, _ = w.fsdp_post_all_gather((codes,), (scale,), BF)
    x = torch.randn(4, 512, 2048, device=DEV, dtype=BF, requires_grad=True)
    y = lin(x)
    assert bits_equal(F.linear(x, packed), y)
    y.float().square().mean().backward()
    assert x.grad.shape == x.shape and lin.weight.grad.shape == (5632, 2048)
    ref = F.linear(x.detach(), w._data)
    rel = (y.detach().float() - ref.float()).norm() / ref.float().norm()
    assert 0.05 < rel < 0.8, rel
    opt = torch.optim.AdamW(lin.parameters(), lr=1e-3)
    before = w._data.clone()
    opt.step()
    assert isinstance(lin.weight, bn.BitNetTrainingLinearWeight) and not torch.equal(lin.weight._data, before)


def test_bitnet_training_steps_learn():
    torch.manual_seed(0)
    model = bn.convert_bitnet(nn.Sequential(nn.Linear(256, 512), nn.GELU(), nn.Linear(512, 256)).to(BF).to(DEV))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    x = torch.randn(512, 256, device=DEV, dtype=BF)
    tgt = torch.randn(512, 256, device=DEV, dtype=BF)
    losses = []
    for _ in range(30):
        loss = (model(x).float() - tgt.float()).square().mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert losses[-1] < losses[0] * 0.95, losses


def test_fsdp2_on_the_device_world1_rccl():
    """fully_shard drives the FSDP2 hooks with DEVICE shards: abs-mean kernel, all_reduce(AVG) over RCCL, fused quantize + 2-bit pack,
    all-gather of the codes, packed forward.  One rank (the GPU box has one card); the 2-rank case runs over gloo on the CPU."""
    import os

    import torch.distributed as dist
    from torch.distributed.fsdp import fully_shard

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(256, 512, bias=False), nn.Linear(512, 128, bias=False)).to(BF).to(DEV)
        single = bn.convert_bitnet(nn.Sequential(nn.Linear(256, 512, bias=False), nn.Linear(512, 128, bias=False)).to(BF).to(DEV))
        single.load_state_dict(model.state_dict())
        bn.convert_bitnet(model)
        for m in model:
            fully_shard(m)
        fully_shard(model)
        x = torch.randn(64, 256, device=DEV, dtype=BF)
        y = model(x)
        assert bits_equal(y, single(x))                      # 2-bit all-gathered weights reproduce the training weight's forward
        y.float().square().mean().backward()
        single(x).float().square().mean().backward()
        g_sharded = model[0].weight.grad.to_local()
        g_sharded = g_sharded._data if hasattr(g_sharded, "_data") else g_sharded
        g_single = single[0].weight.grad
        g_single = g_single._data if hasattr(g_single, "_data") else g_single
        torch.testing.assert_close(g_sharded.float(), g_single.float(), rtol=2e-2, atol=1e-3)
    finally:
        dist.destroy_process_group()
