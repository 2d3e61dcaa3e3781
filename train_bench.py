#!/usr/bin/env python3
"""Llama INT8 mixed-precision pre-training step throughput (tok/s) -- BASELINE.json configs[3] / configs[4].

Restates the harness part of the reference's llm_pretrain.py (model build :100-124, DDP wrap :126-132, AdamW
:57-59,144, loss :30-36, tok/s = batch_size * seq_len * steps / dt :229-234) around this repository's
`subclasses.convert_mixed_precision` and `gn_kernels` ops; synthetic tokens stand in for the dataset and the weights
are random-init (no network here).  The Llama architecture itself comes from `transformers` (as in the reference).

  python train_bench.py --model 470m --quantize mixed_precision                       # configs[3], 1 GPU
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train_bench.py --model 1b --batch_size 16
                                                                                      # configs[4]: global batch 16, DDP over RCCL
"Llama2-1B" is not defined in the reference tree: this harness ASSUMES hidden 2048, intermediate
5632, 22 layers, 32 heads (~1.1 B parameters, the TinyLlama shape) and says so in its output.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]

MODELS = {
    # mini_llamas/Llama-2-470m/config.json:5-10 of the reference
    "470m": dict(hidden_size=1024, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=24, num_key_value_heads=16),
    # assumption, see module docstring
    "1b": dict(hidden_size=2048, intermediate_size=5632, num_attention_heads=32, num_hidden_layers=22, num_key_value_heads=32),
    "tiny": dict(hidden_size=256, intermediate_size=512, num_attention_heads=4, num_hidden_layers=2, num_key_value_heads=4),
}


def build_model(name, seq_len, device, causal_flag=True, sdpa="mem_efficient"):
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama.modeling_llama import LlamaRMSNorm

    cfg = LlamaConfig(vocab_size=32000, rms_norm_eps=1e-5, max_position_embeddings=seq_len, use_cache=False, **MODELS[name])
    if causal_flag:
        # Attention is not part of the scaled-matmul path, but it was 40 % of the compiled step as the reference's harness builds it:
        #  * while dynamo traces, transformers materialises the causal mask and calls SDPA with attn_mask=mask, is_causal=False
        #    (masking_utils: the is_causal shortcut is skipped under tracing): 1.95 ms forward / 5.0 ms backward per Llama-1B layer on
        #    this stack instead of 0.8 / 4.0 ms with the flag.  Whole unpadded sequences need no mask tensor: same SDPA function,
        #    registered under a name whose mask builder returns None;
        #  * the compiled graph always takes SDPA's flash backend, whose backward (aotriton) is 1.7x slower here than the
        #    memory-efficient backend's (tools/sdpa_probe2.py: fwd+bwd 4.75 vs 2.75 ms per layer, identical outputs; the global
        #    enable_*_sdp switches are not consulted while tracing, a `sdpa_kernel` context inside the traced function is).
        # Same setting for bf16 and INT8 runs.
        from torch.nn.attention import SDPBackend, sdpa_kernel
        from transformers import AttentionInterface, AttentionMaskInterface
        from transformers.integrations.sdpa_attention import sdpa_attention_forward

        on_gpu = torch.device(device).type == "cuda"
        backend = {"mem_efficient": SDPBackend.EFFICIENT_ATTENTION, "flash": SDPBackend.FLASH_ATTENTION}.get(sdpa) if on_gpu else None

        def sdpa_causal_flag(module, query, key, value, attention_mask, **kwargs):
            if backend is None:
                return sdpa_attention_forward(module, query, key, value, None, **kwargs)
            with sdpa_kernel(backend):
                return sdpa_attention_forward(module, query, key, value, None, **kwargs)

        impl = f"sdpa_causal_flag_{sdpa if backend is not None else 'any'}"      # one registry entry per backend choice
        AttentionInterface.register(impl, sdpa_causal_flag)
        AttentionMaskInterface.register(impl, lambda *a, **k: None)
        cfg._attn_implementation = impl
    model = LlamaForCausalLM(cfg)
    for m in model.modules():   # llm_pretrain.py:116-118: RoPE cache stays fp32
        if isinstance(m, (nn.Linear, nn.Embedding, LlamaRMSNorm)):
            m.bfloat16()
    return model.to(device)


def get_loss(model, tokens, labels):   # llm_pretrain.py:30-36
    logits = model(tokens).logits.float()
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.long().view(-1))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="470m", choices=list(MODELS))
    ap.add_argument("--quantize", default="mixed_precision", choices=["none", "mixed_precision", "int8", "bitnet"])
    ap.add_argument("--quantize_kwargs", default="{}", help='JSON, as llm_pretrain.py:45: MixedPrecisionConfig / Int8QTConfig fields, e.g. \'{"stochastic_rounding": true}\'')
    ap.add_argument("--batch_size", type=int, default=4, help="GLOBAL batch (split over ranks as llm_pretrain.py:93,151)")
    ap.add_argument("--seq_len", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--weight_decay", type=float, default=1e-2)
    ap.add_argument("--activation_checkpointing", action="store_true")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--data", default="uniform", choices=["uniform", "markov"], help="uniform random tokens (throughput runs) or a learnable Markov stream (loss-curve runs)")
    ap.add_argument("--log_losses", action="store_true", help="record the loss of every step (adds one host sync per step: not for throughput numbers)")
    ap.add_argument("--no_quantize_cache", action="store_true", help="A/B: re-quantize a tensor every time it is used (subclasses.mixed_precision.set_quantize_plan)")
    ap.add_argument("--no_fused_quantize", action="store_true", help="A/B: never use the fused row-wise + column-absmax kernels")
    ap.add_argument("--compile", action="store_true", help="loss_fn = torch.compile(get_loss), as llm_pretrain.py:184 (the gn_kernels ops stay opaque nodes)")
    ap.add_argument("--compile_backend", default="inductor", help="torch.compile backend (tests use aot_eager on the CPU)")
    ap.add_argument("--compile_mode", default=None, help='torch.compile mode, e.g. "reduce-overhead" (forward and backward replayed as HIP graphs: the gn_kernels ops '
                    "allocate through torch's caching allocator and launch on the current stream, so they capture)")
    ap.add_argument("--hf_causal_mask", action="store_true", help="A/B: transformers' default attention call (a materialised causal mask while dynamo traces) "
                    "instead of SDPA's is_causal flag; same for bf16 and INT8 runs")
    ap.add_argument("--sdpa", default="mem_efficient", choices=["default", "flash", "mem_efficient"],
                    help="SDPA backend of the model's attention on the GPU (not part of the scaled-matmul path; the same for bf16 and INT8 runs): "
                    "see build_model")
    ap.add_argument("--ddp_bucket_mb", type=int, default=128, help="DDP gradient bucket size.  xGMI is point-to-point (7 links per GPU): RCCL's rings reach their "
                    "per-link rate only on large messages, so the bf16 gradients of a 1B model (2.5 GB) go out as ~20 buckets of 128 MB instead of torch's default "
                    "100 x 25 MB; the first bucket stays 1 MB (torch's default), so the all-reduce still starts with the first gradients of backward")
    return ap.parse_args(argv)


def run(args):
    """One measured training run; returns the result dict on rank 0 (None elsewhere).  Joins an already initialised process group
    (bench.py calls this in-process after its own timed region) or creates one from the torchrun environment."""
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    on_gpu = args.device == "cuda"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(local_rank)
    own_group = world > 1 and not dist.is_initialized()
    if own_group:
        dist.init_process_group(backend="nccl" if on_gpu else "gloo", **({"device_id": dev} if on_gpu else {}))
    assert args.batch_size % world == 0
    bsize = args.batch_size // world
    torch.manual_seed(1234 + rank)

    model = build_model(args.model, args.seq_len, dev, causal_flag=not args.hf_causal_mask, sdpa=args.sdpa)
    if args.activation_checkpointing:
        model.gradient_checkpointing_enable()
    if args.no_quantize_cache or args.no_fused_quantize:
        from subclasses import mixed_precision as _mp

        _mp.set_quantize_plan(cache=not args.no_quantize_cache, fused=not args.no_fused_quantize)
    qkw = json.loads(args.quantize_kwargs)
    if args.quantize == "mixed_precision":
        from subclasses import MixedPrecisionConfig, convert_mixed_precision

        convert_mixed_precision(model.model, config=MixedPrecisionConfig(**qkw))   # llm_pretrain.py:122 (lm_head stays bf16 unless --quantize_lm_head)
    elif args.quantize == "int8":
        from subclasses import Int8QTConfig, convert_int8_quantized_training

        convert_int8_quantized_training(model.model, config=Int8QTConfig(**qkw))
    elif args.quantize == "bitnet":
        from subclasses import convert_bitnet

        convert_bitnet(model.model)                   # train_utils.py's extra RMSNorm in front of every Linear is model surgery, not part of this path
    n_params = sum(p.numel() for p in model.parameters())
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP

        # gradient_as_bucket_view: .grad aliases the bucket, which saves one copy of every gradient per step and 2 bytes per parameter
        model = DDP(model, device_ids=[local_rank] if on_gpu else None, bucket_cap_mb=args.ddp_bucket_mb, gradient_as_bucket_view=True)
    optim = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=args.weight_decay, fused=on_gpu and args.quantize != "int8")

    # --data markov: a LEARNABLE synthetic stream (next token = perm[token] with probability 0.9, else uniform), so that loss curves of
    # the quantized and bf16 runs can be compared while the loss actually falls; uniform random tokens only ever reach log(32000).
    data_gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    perm = torch.randperm(32000, generator=torch.Generator().manual_seed(99)).to(dev)

    def batch():
        if args.data == "uniform":
            t = torch.randint(0, 32000, (bsize, args.seq_len + 1), device=dev)
        else:
            noise = torch.randint(0, 32000, (bsize, args.seq_len + 1), device=dev, generator=data_gen)
            keep = torch.rand((bsize, args.seq_len + 1), device=dev, generator=data_gen) < 0.9
            t = noise.clone()
            for i in range(1, args.seq_len + 1):                       # short python loop over positions (data only, outside the model)
                t[:, i] = torch.where(keep[:, i], perm[t[:, i - 1]], noise[:, i])
        return t[:, :-1].contiguous(), t[:, 1:].contiguous()

    loss_fn = torch.compile(get_loss, backend=args.compile_backend, mode=args.compile_mode) if args.compile else get_loss     # llm_pretrain.py:184

    def step():
        tokens, labels = batch()
        loss = loss_fn(model, tokens, labels)
        loss.backward()
        optim.step()
        optim.zero_grad()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(step().item())
    sync()
    t0 = time.perf_counter()
    curve = []
    for _ in range(args.steps):
        last = step()
        if args.log_losses:
            curve.append(round(last.item(), 4))
    sync()
    dt = time.perf_counter() - t0
    losses.append(last.item())
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    out = None
    if rank == 0:
        toks = args.batch_size * args.seq_len * args.steps
        lin = sum(m.in_features * m.out_features for m in model.modules() if isinstance(m, nn.Linear) and m.out_features != 32000)
        out = {
            "metric": "Llama INT8 mixed-precision training tok/s", "value": round(toks / dt, 1), "unit": "tok/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "dtype": {"none": "bf16", "bitnet": "bf16 weights/activations; int8 x ternary MFMA in the Linear fwd, bf16 bwd"}.get(
                args.quantize, "bf16 weights/activations; int8 MFMA in the Linear fwd / grad-input / grad-weight"),
            "data": "synthetic tokens, random-init weights",
            "config": {"workload": f"llama-{args.model} {'(ASSUMED 1.1B shape: 2048/5632/22/32)' if args.model == '1b' else ''} global_batch {args.batch_size} x seq_len {args.seq_len}, "
                                   f"quantize={args.quantize}{' ' + args.quantize_kwargs if qkw else ''}, {'torch.compile(get_loss)' if args.compile else 'eager'}, AdamW, {'DDP x%d (RCCL all-reduce)' % world if world > 1 else 'single GPU'}",
                       "params": n_params, "body_linear_weights": lin},
            "body_linear_TOPS": round(6 * args.batch_size * args.seq_len * lin * args.steps / dt * 1e-12, 1),
            "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
            **({"loss_curve": curve, "data": f"synthetic {args.data} tokens, random-init weights"} if args.log_losses else {}),
        }
        if on_gpu:
            out["max_mem_GiB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    if world > 1:
        dist.barrier()
    if own_group:
        dist.destroy_process_group()
    return out


def main():
    out = run(parse_args())
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
