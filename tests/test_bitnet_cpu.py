"""CPU checks of the BitNet path: the numpy oracle (oracle/bitnet_ref.py) and the host logic of subclasses/bitnet.py against
the goldens generated from the reference (tools/gen_golden_bitnet.py -> tests/golden/g8_bitnet.npz); FSDP2 over gloo.

Bars: quantize / pack / unpack bit-exact; the abs-mean scale to 1e-6 relative (an fp32 sum whose order is the library's);
the Linear forward bit-exact (integer product, one fp32 scale multiply chain); backward products are bf16 library matmuls
(`grad_output @ w_i8.to(bf16)`), compared at the reference's own bf16 tolerance (rtol 1.6e-2) because their summation order
belongs to the BLAS."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F
from torch import nn

from oracle import bitnet_ref
from subclasses import bitnet as bn
from tests._util import bits_equal, load, t_from_bits

BF = torch.bfloat16
WEIGHTS = {"bf16_40x72": BF, "bf16_5x12": BF, "f32_7x20": torch.float32, "bf16_256x384": BF}


def _w(g, name):
    return t_from_bits(g[f"w_{name}"], WEIGHTS.get(name, BF))


@pytest.mark.parametrize("name", list(WEIGHTS))
def test_scale_and_ternary_quantization_match_the_reference(name):
    g = load("g8_bitnet.npz")
    w, want_s, want_q = _w(g, name), g[f"s_{name}"], g[f"q_{name}"]
    s = bitnet_ref.get_bitnet_scale(w.float().numpy())
    assert abs(float(s) - float(want_s)) <= 1e-6 * float(want_s)
    assert np.array_equal(bitnet_ref.quantize_bitnet_weight(w.float().numpy(), np.float32(want_s)), want_q)
    assert np.array_equal(bitnet_ref.pack_i2(want_q), g[f"p_{name}"])
    assert np.array_equal(bitnet_ref.unpack_i2(g[f"p_{name}"]), want_q)           # ternary data survives the 2-bit round trip
    # host path of the product module (ATen on CPU tensors)
    hs = bn.get_bitnet_scale(w)
    assert hs.dtype is torch.float32 and hs.shape == ()
    assert abs(hs.item() - float(want_s)) <= 1e-6 * float(want_s)
    assert np.array_equal(bn.quantize_bitnet_weight(w, torch.tensor(float(want_s))).numpy(), want_q)
    assert np.array_equal(bn._pack_i2_in_i8(torch.from_numpy(want_q)).numpy(), g[f"p_{name}"])
    assert np.array_equal(bn._quantize_and_pack(w, torch.tensor(float(want_s))).numpy(), g[f"p_{name}"])


def test_rounding_ties_and_the_divisor_floor():
    g = load("g8_bitnet.npz")
    for tag, dt in (("ties", BF), ("tiny", torch.float32)):
        w = t_from_bits(g[f"w_{tag}"], dt)
        s = np.float32(g[f"s_{tag}"])
        assert np.array_equal(bitnet_ref.quantize_bitnet_weight(w.float().numpy(), s), g[f"q_{tag}"])
        assert np.array_equal(bn.quantize_bitnet_weight(w, torch.tensor(float(s))).numpy(), g[f"q_{tag}"])
    q = g["q_ties"].reshape(-1)[:16].tolist()
    assert q[:6] == [0, 0, 1, -1, 1, -1] and q[8:] == [1, -1, 0, 1, -1, 0, 0, 0]   # 0.5 -> 0, 1.5 -> 2 -> clip 1: half to even


def test_pack_and_unpack_arbitrary_bytes_match_the_reference():
    g = load("g8_bitnet.npz")
    assert np.array_equal(bitnet_ref.pack_i2(g["pack_in"]), g["pack_out"])
    assert np.array_equal(bitnet_ref.unpack_i2(g["unpack_in"]), g["unpack_out"])
    assert np.array_equal(bn._pack_i2_in_i8(torch.from_numpy(g["pack_in"])).numpy(), g["pack_out"])
    assert np.array_equal(bn._unpack_i2_in_i8(torch.from_numpy(g["unpack_in"])).numpy(), g["unpack_out"])


def _close(got, want):
    torch.testing.assert_close(got.float(), want.float(), rtol=1.6e-2, atol=2e-3)


def test_training_linear_matches_the_reference(oracle_cpu_ops):
    g = load("g8_bitnet.npz")
    lin = nn.Linear(256, 128, bias=True).to(BF)
    with torch.no_grad():
        lin.weight.copy_(t_from_bits(g["train_W"], BF))
        lin.bias.copy_(t_from_bits(g["train_b"], BF))
    bn.convert_bitnet(lin)
    assert isinstance(lin.weight, bn.BitNetTrainingLinearWeight) and isinstance(lin.weight, nn.Parameter)
    x = t_from_bits(g["train_x"], BF).requires_grad_()
    y = lin(x)
    y.backward(t_from_bits(g["train_gy"], BF))
    assert bits_equal(y, t_from_bits(g["train_y"], BF))
    _close(x.grad, t_from_bits(g["train_dx"], BF))
    gw = lin.weight.grad
    _close(gw._data if isinstance(gw, bn.BitNetTrainingLinearWeight) else gw, t_from_bits(g["train_dW"], BF))
    _close(lin.bias.grad, t_from_bits(g["train_db"], BF))


def test_packed_2bit_linear_matches_the_reference(oracle_cpu_ops):
    g = load("g8_bitnet.npz")
    w = bn.BitNetPacked2bitLinearWeight(torch.from_numpy(g["packed_i2"]), t_from_bits(g["packed_scale"], BF).reshape(()))
    assert w.shape == (128, 256) and w.dtype is BF
    assert bits_equal(w.dequantize(), t_from_bits(g["packed_deq"], BF))
    x = t_from_bits(g["packed_x"], BF).requires_grad_()
    y = F.linear(x, w)                              # no bias slot at all: backward must cope
    y.backward(t_from_bits(g["packed_gy"], BF))
    assert bits_equal(y, t_from_bits(g["packed_y"], BF))
    _close(x.grad, t_from_bits(g["packed_dx"], BF))
    assert isinstance(w.detach(), bn.BitNetPacked2bitLinearWeight) and isinstance(w.clone(), bn.BitNetPacked2bitLinearWeight)
    with pytest.raises(NotImplementedError):
        w + 1


def test_dispatch_table_optimizer_and_state_dict(oracle_cpu_ops):
    torch.manual_seed(5)
    model = bn.convert_bitnet(nn.Sequential(nn.Linear(32, 64), nn.ReLU(), nn.Linear(64, 16)).to(BF))
    w = model[0].weight
    for view in (w.detach(), w.t(), w.clone(), w.view(-1), w[:8]):
        assert isinstance(view, bn.BitNetTrainingLinearWeight)
    assert not isinstance(w + 1, bn.BitNetTrainingLinearWeight)
    assert w.__tensor_flatten__() == (["_data"], [])
    sd = copy.deepcopy(model.state_dict())
    before = w._data.clone()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    model(torch.randn(4, 32, dtype=BF)).square().mean().backward()
    opt.step()
    assert isinstance(model[0].weight, bn.BitNetTrainingLinearWeight) and not torch.equal(model[0].weight._data, before)
    model.load_state_dict(sd)
    assert torch.equal(model[0].weight._data, before)
    # quantization error of the ternary forward stays what 1.58 bits allow
    ref = nn.Linear(256, 64, bias=False).to(BF)
    tern = bn.convert_bitnet(copy.deepcopy(ref))
    xs = torch.randn(32, 256, dtype=BF)
    rel = (tern(xs).float() - ref(xs).float()).norm() / ref(xs).float().norm()
    assert 0.05 < rel < 0.8, rel


# ------------------------------------------------------------------------------- FSDP2 over gloo
def _fsdp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root / "quantized-training_amd"), str(root)]
    import gn_kernels  # noqa: F401
    import oracle
    from subclasses import bitnet as bnw
    from torch.distributed.fsdp import fully_shard

    impl = torch.library.Library("gn_kernels", "IMPL")
    impl.impl("scaled_mm", lambda A, B, sa, sb: oracle.scaled_mm(A, B, sa, sb), "CPU")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(64, 128, bias=False), nn.Linear(128, 32, bias=False)).to(BF)
        single = bnw.convert_bitnet(copy.deepcopy(model))
        bnw.convert_bitnet(model)
        # the hook pair by hand on this rank's half of the first weight: scale = mean of shard means, bytes = packed codes
        full = single[0].weight._data
        shard = bnw.BitNetTrainingLinearWeight(full.chunk(world)[rank].contiguous())
        (codes,), (scale,) = shard.fsdp_pre_all_gather(None)
        want_scale = bnw.get_bitnet_scale(full)
        ok = abs(scale.item() - want_scale.item()) <= 1e-6 * want_scale.item()
        ok &= torch.equal(codes, bnw._pack_i2_in_i8(bnw.quantize_bitnet_weight(shard._data, scale)))
        gathered = [torch.empty_like(codes) for _ in range(world)]
        dist.all_gather(gathered, codes)
        packed, _ = shard.fsdp_post_all_gather((torch.cat(gathered),), (scale,), BF)
        ok &= isinstance(packed, bnw.BitNetPacked2bitLinearWeight) and packed.shape == full.shape
        ok &= torch.equal(bnw._unpack_i2_in_i8(packed.int_data), bnw.quantize_bitnet_weight(full, scale))
        # the real thing: fully_shard drives the hooks; the all-gather moves 2-bit codes, outputs equal the unsharded model's
        for m in model:
            fully_shard(m)
        fully_shard(model)
        x = torch.randn(8, 64, generator=torch.Generator().manual_seed(1)).to(BF)
        y = model(x)
        ok &= torch.equal(y, single(x))
        y.float().square().mean().backward()
        single(x).float().square().mean().backward()
        g_full = single[0].weight.grad
        g_full = g_full._data if hasattr(g_full, "_data") else g_full
        g_loc = model[0].weight.grad.to_local()
        g_loc = g_loc._data if hasattr(g_loc, "_data") else g_loc
        ok &= torch.allclose(g_loc.float(), g_full.chunk(world)[rank].float(), rtol=2e-2, atol=1e-3)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_fsdp2_gloo_world2_all_gathers_2bit_weights():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_fsdp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
