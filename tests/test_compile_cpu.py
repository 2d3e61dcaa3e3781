"""torch.compile over the INT8 mixed-precision subclass path, on CPU (reference: llm_pretrain.py:184 `torch.compile(get_loss)`,
:134-139 `layer.compile()`; llm_finetune.py compiles with fullgraph=True).

Host tensors are routed through the DEVICE branch of the subclasses (`gn_kernels.quantize.on_device` patched to True) with the ORACLE
registered as the ops' CPU backend (tests/conftest.py), so what dynamo / AOTAutograd trace here is exactly what they trace on the GPU:
the F.linear hijack, the quantization plan (fused row-wise + column-absmax ops, column-wise-given ops), the custom ops' Meta kernels.
`backend="aot_eager"` runs the traced forward + backward graphs with the real op implementations: results must equal eager mode's
bit for bit (the oracle is deterministic)."""
import copy

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from gn_kernels import quantize as qmod
from subclasses import MixedPrecisionConfig, convert_mixed_precision
from subclasses import mixed_precision as mp_mod
from tests._util import bits_equal

BF = torch.bfloat16


@pytest.fixture
def device_branch_on_cpu(oracle_cpu_ops, monkeypatch):
    monkeypatch.setattr(qmod, "on_device", lambda t: True)
    torch._dynamo.reset()
    yield
    torch._dynamo.reset()
    mp_mod.clear_quantize_cache()


def _tiny_llama(layers=2):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=4,
                      num_hidden_layers=layers, max_position_embeddings=32, use_cache=False, rms_norm_eps=1e-5)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).to(BF)


def get_loss(model, tokens, labels):                      # llm_pretrain.py:30-36
    logits = model(tokens).logits.float()
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.long().view(-1))


def _grads(model):
    out = {}
    for n, p in model.named_parameters():
        g = p.grad
        out[n] = (g._data if hasattr(g, "_data") else g).detach().clone()
    return out


@pytest.mark.parametrize("fused_min", [0, 1 << 62], ids=["fused-pair", "separate-passes"])
def test_compiled_llama_step_equals_eager_bitwise(device_branch_on_cpu, monkeypatch, fused_min):
    monkeypatch.setattr(mp_mod, "_FUSED_MIN_NUMEL", fused_min)
    model = _tiny_llama()
    convert_mixed_precision(model.model)
    ref = copy.deepcopy(model)
    torch.manual_seed(1)
    tokens = torch.randint(0, 128, (2, 16))
    labels = torch.randint(0, 128, (2, 16))

    loss_e = get_loss(ref, tokens, labels)
    loss_e.backward()

    loss_fn = torch.compile(get_loss, backend="aot_eager", fullgraph=True)     # fullgraph: no graph break anywhere, the Linears included
    loss_c = loss_fn(model, tokens, labels)
    loss_c.backward()

    assert bits_equal(loss_c.detach(), loss_e.detach())
    ge, gc = _grads(ref), _grads(model)
    assert ge.keys() == gc.keys()
    for name in ge:
        assert bits_equal(gc[name].contiguous(), ge[name].contiguous()), name


def test_compiled_graph_merges_duplicate_quantizations_and_keeps_stochastic_ones_apart(device_branch_on_cpu, monkeypatch):
    """q/k/v (gate/up) share their input: the compiled graph quantizes it once (what the eager-mode cache does by hand), but under
    stochastic rounding every call must keep its own draw -- the `rng` argument makes the calls differ."""
    import oracle

    calls = []
    real = oracle.quantize_int8

    def spy(x, sr=False, **kw):
        calls.append((tuple(x.shape), kw.get("dim"), sr, kw.get("seed")))
        return real(x, sr, **kw)

    monkeypatch.setattr(oracle, "quantize_int8", spy)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v = (nn.Linear(32, 32, bias=False) for _ in range(3))

        def forward(self, x):
            h = x * 2
            return self.q(h) + self.k(h) + self.v(h)

    def run(sr, compiled):
        torch.manual_seed(0)
        blk = convert_mixed_precision(Block().to(BF), config=MixedPrecisionConfig(stochastic_rounding=sr, grad_input=False, grad_weight=False))
        # the min-cut partitioner of a TRAINING graph is where AOTAutograd runs its common-subexpression pass (as under inductor)
        fn = torch.compile(blk, backend="aot_eager_decomp_partition", fullgraph=True) if compiled else blk
        calls.clear()
        mp_mod.clear_quantize_cache()
        torch.manual_seed(5)
        x = torch.randn(8, 32, dtype=BF, requires_grad=True)
        y = fn(x)
        y.sum().backward()
        return y.detach(), list(calls)

    x_shape = (8, 32)
    y_e, c_e = run(False, False)
    y_c, c_c = run(False, True)
    assert bits_equal(y_e, y_c)
    assert sum(1 for c in c_e if c[0] == x_shape) == 1                 # eager: the identity cache serves k and v
    assert sum(1 for c in c_c if c[0] == x_shape) == 1                 # compiled: the compiler merged the three identical nodes
    y_e, c_e = run(True, False)
    y_c, c_c = run(True, True)
    assert sum(1 for c in c_e if c[0] == x_shape) == 3                 # stochastic rounding: never cached ...
    assert sum(1 for c in c_c if c[0] == x_shape) == 3                 # ... and never merged
    assert len({c[3] for c in c_c}) == len(c_c) == 6                   # six calls (3 x h, 3 x W), six different Philox keys
    assert bits_equal(y_e, y_c)                                        # same torch seed -> same keys in the same order -> same bits


def test_eager_cache_is_identity_scoped(device_branch_on_cpu):
    """Eager-mode cache: keyed by tensor identity + version, only for tensors inside an autograd graph, never for
    inference tensors, and it holds no strong reference to the activation."""
    import gc
    import weakref

    lin = convert_mixed_precision(nn.Linear(32, 16, bias=False).to(BF))
    mp_mod.clear_quantize_cache()
    x = torch.randn(8, 32, dtype=BF, requires_grad=True)
    h = x * 1                                  # has a grad_fn: cacheable
    y1 = lin(h)
    assert len(mp_mod._cache) == 1
    ref = weakref.ref(h)
    y2 = lin(h)
    assert bits_equal(y1, y2)
    with torch.no_grad():
        h.add_(1)                              # bumps the version: the entry must not be served
    y3 = lin(h)
    assert not bits_equal(y1.detach(), y3.detach())
    del h, y1, y2, y3
    gc.collect()
    assert ref() is None and len(mp_mod._cache) == 0      # the cache did not keep the activation alive, and its entry went with it
    # leaves (no grad_fn) are never cached: a user may rewrite them through .data without a version bump
    leaf = torch.randn(8, 32, dtype=BF)
    a = lin(leaf)
    leaf.data.copy_(torch.randn(8, 32, dtype=BF))
    b = lin(leaf)
    assert len(mp_mod._cache) == 0 and not bits_equal(a.detach(), b.detach())
    with torch.inference_mode():               # inference tensors have no version counter
        t = torch.randn(8, 32, dtype=BF) * 1
        lin(t)
    assert len(mp_mod._cache) == 0


@pytest.mark.parametrize("activation", ["none", "int8"])
def test_compiled_int8_quantized_training_equals_eager(device_branch_on_cpu, activation):
    """Int8LinearWeight (reference subclasses/int8.py:144-207; llm_finetune.py compiles the model with fullgraph=True): weight-only and
    dynamic-INT8-activation forwards trace without a graph break and reproduce eager mode bit for bit, gradients included."""
    from subclasses import Int8QTConfig, convert_int8_quantized_training

    model = _tiny_llama()
    convert_int8_quantized_training(model.model, config=Int8QTConfig(activation=activation))
    ref = copy.deepcopy(model)
    torch.manual_seed(1)
    tokens = torch.randint(0, 128, (2, 16))
    labels = torch.randint(0, 128, (2, 16))
    loss_e = get_loss(ref, tokens, labels)
    loss_e.backward()
    loss_c = torch.compile(get_loss, backend="aot_eager", fullgraph=True)(model, tokens, labels)
    loss_c.backward()
    assert bits_equal(loss_c.detach(), loss_e.detach())
    for (n, pe), (_, pc) in zip(ref.named_parameters(), model.named_parameters()):
        assert (pe.grad is None) == (pc.grad is None), n
        if pe.grad is not None:
            assert bits_equal(pc.grad.contiguous(), pe.grad.contiguous()), n
