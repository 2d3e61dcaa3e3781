"""torch.compile (inductor) over the INT8 mixed-precision path on the GPU -- how the reference's harness runs it
(llm_pretrain.py:184 `torch.compile(get_loss)`, :134-139 `layer.compile()`, llm_finetune.py fullgraph=True).
The gn_kernels ops are opaque nodes with Meta kernels; every plan decision is taken from shapes and strides."""
import copy

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from subclasses import MixedPrecisionConfig, convert_mixed_precision
from subclasses import mixed_precision as mp_mod
from tests._util import bits_equal

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def get_loss(model, tokens, labels):                      # llm_pretrain.py:30-36
    logits = model(tokens).logits.float()
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.long().view(-1))


@pytest.fixture(autouse=True)
def _fresh_dynamo():
    torch._dynamo.reset()
    yield
    torch._dynamo.reset()


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("fused_min", [0, 1 << 62], ids=["fused-pair", "separate-passes"])
def test_compiled_linear_fullgraph_is_bit_identical_to_eager(monkeypatch, bias, fused_min):
    """One converted Linear, forward + backward, fullgraph=True: the graph holds only the custom ops and views, so compiled == eager bit for bit."""
    monkeypatch.setattr(mp_mod, "_FUSED_MIN_NUMEL", fused_min)
    torch.manual_seed(0)
    lin = nn.Linear(256, 512, bias=bias).to(BF).to(DEV)
    convert_mixed_precision(lin)
    ref = copy.deepcopy(lin)
    x = torch.randn(4, 64, 256, device=DEV, dtype=BF, requires_grad=True)
    x2 = x.detach().clone().requires_grad_()
    gy = torch.randn(4, 64, 512, device=DEV, dtype=BF)
    ref(x2).backward(gy)
    clin = torch.compile(lin, fullgraph=True)
    clin(x).backward(gy)
    assert bits_equal(x.grad, x2.grad)
    gw, gw2 = lin.weight.grad, ref.weight.grad
    assert bits_equal(getattr(gw, "_data", gw).contiguous(), getattr(gw2, "_data", gw2).contiguous())
    with torch.no_grad():
        assert bits_equal(clin(x), ref(x2))
    if bias:
        assert bits_equal(lin.bias.grad, ref.bias.grad)


def test_compiled_two_layer_llama_loss_matches_eager_with_zero_graph_breaks():
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_attention_heads=4, num_key_value_heads=4,
                      num_hidden_layers=2, max_position_embeddings=128, use_cache=False, rms_norm_eps=1e-5)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).to(BF).to(DEV)
    convert_mixed_precision(model.model)
    ref = copy.deepcopy(model)
    tokens = torch.randint(0, 1024, (4, 128), device=DEV)
    labels = torch.randint(0, 1024, (4, 128), device=DEV)
    loss_e = get_loss(ref, tokens, labels)
    loss_e.backward()
    loss_fn = torch.compile(get_loss, fullgraph=True)          # fullgraph=True: ANY graph break (the Linears included) raises
    loss_c = loss_fn(model, tokens, labels)
    loss_c.backward()
    torch.cuda.synchronize()
    # stated tolerance: |loss difference| <= 1e-2; the compiled graph fuses the bf16 glue around the Linears differently
    assert abs(loss_c.item() - loss_e.item()) <= 1e-2, (loss_c.item(), loss_e.item())
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        g, h = getattr(p.grad, "_data", p.grad).float(), getattr(q.grad, "_data", q.grad).float()
        rel = (g - h).norm() / h.norm().clamp_min(1e-12)
        assert rel < 0.1, (n, rel.item())     # int8 re-quantization of slightly different bf16 activations: noise-sized, not bitwise
    # a second step reuses the compiled graph (no recompile): the optimizer rewrote the weights in place
    before = torch._dynamo.utils.counters["stats"]["unique_graphs"]
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    opt.step()
    opt.zero_grad()
    loss2 = loss_fn(model, tokens, labels)
    loss2.backward()
    torch.cuda.synchronize()
    assert torch._dynamo.utils.counters["stats"]["unique_graphs"] == before
    assert loss2.item() < loss_c.item()


def test_compiled_stochastic_rounding_draws_differ_between_calls_and_steps():
    torch.manual_seed(0)
    lin = nn.Linear(256, 256, bias=False).to(BF).to(DEV)
    convert_mixed_precision(lin, config=MixedPrecisionConfig(stochastic_rounding=True, grad_input=False, grad_weight=False))
    clin = torch.compile(lin, fullgraph=True)
    x = torch.randn(64, 256, device=DEV, dtype=BF)
    with torch.no_grad():
        a, b = clin(x), clin(x)                       # same compiled graph replayed: new noise each time
        torch.manual_seed(7)
        c = clin(x)
        torch.manual_seed(7)
        d = clin(x)
    assert not bits_equal(a, b)
    assert bits_equal(c, d)                           # torch.manual_seed reproduces the draw


@pytest.mark.parametrize("activation", ["none", "int8"])
def test_compiled_int8_quantized_training_linear_matches_eager(activation):
    """Int8LinearWeight (reference subclasses/int8.py:144-207) under inductor, fullgraph=True: the dynamic-INT8 forward is custom ops only and
    must equal eager bit for bit; the weight-only forward dequantizes in generated code (bf16 `weight * scale` then matmul), so it is held to
    the bf16 tolerance of the reference's own check (1e-2)."""
    from subclasses import Int8QTConfig, convert_int8_quantized_training

    torch.manual_seed(0)
    lin = nn.Linear(256, 512, bias=False).to(BF).to(DEV)
    convert_int8_quantized_training(lin, config=Int8QTConfig(activation=activation))
    ref = copy.deepcopy(lin)
    x = torch.randn(4, 64, 256, device=DEV, dtype=BF, requires_grad=True)
    x2 = x.detach().clone().requires_grad_()
    gy = torch.randn(4, 64, 512, device=DEV, dtype=BF)
    ref(x2).backward(gy)
    clin = torch.compile(lin, fullgraph=True)
    clin(x).backward(gy)
    with torch.no_grad():
        out, want = clin(x), ref(x2)
    if activation == "int8":
        assert bits_equal(out, want)
    else:
        torch.testing.assert_close(out.float(), want.float(), rtol=1e-2, atol=1e-2 * want.float().abs().max().item())
    torch.testing.assert_close(x.grad.float(), x2.grad.float(), rtol=1e-2, atol=1e-2 * x2.grad.float().abs().max().item())
    torch.testing.assert_close(lin.weight.grad.float(), ref.weight.grad.float(), rtol=1e-2, atol=1e-2 * ref.weight.grad.float().abs().max().item())


def test_ddp_over_rccl_with_compiled_loss_single_rank():
    """The configuration bench.py's detail_train runs at N > 1 (train_bench.run: DDP with 128 MB buckets as gradient views around the
    converted model, torch.compile(get_loss), fused AdamW), on the one GPU of this box: a one-rank RCCL group.  Two steps; the losses
    equal those of the same model trained without DDP (same compiled graph, the all-reduce of one rank is the identity)."""
    import os
    import sys
    from pathlib import Path

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import train_bench

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29900 + os.getpid() % 100)        # its own port: another test of this process may have used the default
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(5)
        base = train_bench.build_model("tiny", 128, torch.device(DEV))
        convert_mixed_precision(base.model)
        tokens = torch.randint(0, 32000, (4, 128), device=DEV)
        labels = torch.randint(0, 32000, (4, 128), device=DEV)
        losses = {}
        for name in ("plain", "ddp"):
            torch._dynamo.reset()
            model = copy.deepcopy(base)
            if name == "ddp":
                model = DDP(model, device_ids=[0], bucket_cap_mb=128, gradient_as_bucket_view=True)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
            loss_fn = torch.compile(train_bench.get_loss)
            out = []
            for _ in range(2):
                loss = loss_fn(model, tokens, labels)
                loss.backward()
                opt.step()
                opt.zero_grad()
                out.append(loss.item())
            losses[name] = out
        assert losses["ddp"][0] == losses["plain"][0]
        assert abs(losses["ddp"][1] - losses["plain"][1]) <= 1e-2 and losses["ddp"][1] < losses["ddp"][0] + 1e-2
    finally:
        if created:
            dist.destroy_process_group()
