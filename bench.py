#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X scaled low-precision matmul path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1], the configuration the TOPS metric is quoted on; inputs follow
the reference's benchmark_mm.py:60,78-96,125-126): one STEP = the six scaled GEMMs at M=N=K=4096 --
scaled INT8 and scaled FP8 (e4m3), each in the three operand layouts A@B.T, A@B and A.T@B -- through
`gn_kernels.scaled_mm` (op -> C ABI -> hand-written gfx950 kernels), bf16 scales and bf16 output.
Operands are resident in HBM before the timed region.  `value` is the whole-job rate
(2*M*N*K ops per GEMM, the reference's convention benchmark_mm.py:107) over all ranks.

With --gpus N > 1 (launched by torch.distributed.run) every rank runs the same step on its own GPU
("replicas only": a single GEMM is not split across GPUs) -> weak scaling.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (scaled INT8 A@B.T), timed live with HIP events on the launch stream
  cpu_baseline -- the reference's CPU fallback formula (torch._int_mm + dequant scale) on the host cores
  detail       -- per-kernel TOPS of the sweep 1024/2048/4096 and quantize-pass GB/s (outside the timed region)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT / "quantized-training_amd"), str(ROOT)]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_DENSE_8BIT_TOPS = 5033.0      # 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz (MI355X_MICROARCH.md: i8 and MX-fp8 = 2x bf16)
PEAK_HBM_GBS = 8000.0


def to_layout(x, column_major):
    return x.T.contiguous().T if column_major else x.contiguous()


LAYOUTS = {"A@B.T": (False, True), "A@B": (False, False), "A.T@B": (True, False)}


def make_inputs(S, dev):
    """benchmark_mm.py:82-86,125-126"""
    A_i8 = torch.randint(-128, 127, (S, S), dtype=torch.int8, device=dev)
    B_i8 = torch.randint(-128, 127, (S, S), dtype=torch.int8, device=dev)
    A_f8 = torch.randn(S, S, device=dev).to(torch.float8_e4m3fn)
    B_f8 = torch.randn(S, S, device=dev).to(torch.float8_e4m3fn)
    sA = torch.randn(S, 1, device=dev).div(128).bfloat16()
    sB = torch.randn(1, S, device=dev).div(128).bfloat16()
    tsA = torch.randn(S, S // 128, device=dev).div(128).bfloat16()
    tsB = torch.randn(S // 128, S // 128, device=dev).div(128).bfloat16()
    cases = {}
    for name, (acm, bcm) in LAYOUTS.items():
        cases[("int8", name)] = (to_layout(A_i8, acm), to_layout(B_i8, bcm), sA, sB)
        cases[("fp8", name)] = (to_layout(A_f8, acm), to_layout(B_f8, bcm), sA, sB)
        cases[("tile_int8", name)] = (to_layout(A_i8, acm), to_layout(B_i8, bcm), to_layout(tsA, acm), to_layout(tsB, bcm))
        cases[("tile_fp8", name)] = (to_layout(A_f8, acm), to_layout(B_f8, bcm), to_layout(tsA, acm), to_layout(tsB, bcm))
    if S >= 4096:   # BASELINE configs[2]: Jetfire-style 32 x 32 x 32 blocks (the same op with (M/32, K/32) / (K/32, N/32) scales)
        jsA = torch.randn(S // 32, S // 32, device=dev).div(128).bfloat16()
        jsB = torch.randn(S // 32, S // 32, device=dev).div(128).bfloat16()
        for name, (acm, bcm) in LAYOUTS.items():
            cases[("tile_int8_jetfire32", name)] = (to_layout(A_i8, acm), to_layout(B_i8, bcm), jsA, jsB)
            cases[("tile_fp8_jetfire32", name)] = (to_layout(A_f8, acm), to_layout(B_f8, bcm), jsA, jsB)
    return cases


PMC_TRAFFIC_BYTES = (2 * 49_383 + 32_770) * 1024   # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; see roofline.traffic_note (profiles/r03x_pmc_int8_nt_4096_w4.txt:
                                                    # the round's final kernel; r03c, its first build, read 49,398 / 32,768)


PREWARM_MS = 120.0   # tools/ramp_probe.py: after an idle period the six-GEMM step takes ~80 steps (~30 ms) to reach its steady time
                     # (first 20 steps 375-397 us, steps 20-40 355, 40-80 340, then 332 us: the chip's clocks ramp with load)


def prewarm(fn, ms=PREWARM_MS):
    """Run `fn` for `ms` milliseconds of GPU time, untimed: brings the clocks to the state they hold under sustained load, so that what
    follows measures steady-state throughput and not the ramp after an idle period.  Returns (calls, ms)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    calls = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            fn()
        calls += 8
        torch.cuda.synchronize()
    return calls, (time.perf_counter() - t0) * 1e3


_last_gpu_work = [0.0]


def prewarm_if_idle(fn, idle_s=0.05, ms=40.0):
    """The detail tables time many short regions with host work (allocation, the CPU baseline, sub-processes) between them: pre-warm again
    whenever the GPU has been idle for more than `idle_s`."""
    now = time.perf_counter()
    if now - _last_gpu_work[0] > idle_s:
        prewarm(fn, ms)


def time_kernel(fn, iters=30, warm=5):
    """Average launch duration in ms: ONE pair of HIP events on the current (= launch) stream around `iters` back-to-back launches, divided
    by iters -- launch gaps between consecutive kernels included.  Second value: the median over per-launch event pairs, the method of
    rounds 1-2; each pair of event records costs packets of its own (tools/probe_launch.hip: an empty kernel reads 6.0 us that way and
    2.3-2.6 us back to back), so it reads ~3 us above the region average at any kernel size."""
    prewarm_if_idle(fn)
    for _ in range(warm):
        fn()
    start = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        start[i].record()
        fn()
        stop[i].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in zip(start, stop))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    _last_gpu_work[0] = time.perf_counter()
    return a.elapsed_time(b) / iters, ts[len(ts) // 2]


def time_kernel_graph(fn, per_graph=20, replays=6):
    """Average launch duration in ms with the Python / ctypes launch path taken out: `per_graph` calls are captured into a
    HIP graph once and the graph is replayed (the reference's do_bench sees ~3 us CUDA launches; our op costs ~13 us of
    host time per call, which would otherwise be what the 1024 / 2048 sizes measure)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    prewarm(g.replay, 40.0)        # capture is host work: the GPU idled meanwhile
    ts = []
    for _ in range(replays):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / per_graph)
    ts.sort()
    return ts[len(ts) // 2]


def _median_time(fn, budget_s, max_runs=20, warm=2):
    for _ in range(warm):
        fn()
    times = []
    t_end = time.perf_counter() + budget_s
    while (time.perf_counter() < t_end and len(times) < max_runs) or len(times) < 3:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2], len(times)


def cpu_baseline(budget_s=20.0):
    """The reference's CPU path, as BASELINE.md section 4 plans it (configs[0]): quantize_int8 of both operands (subclasses/int8.py:12-26
    in ATen ops), torch._int_mm, fp32 dequant scale -> bf16 (benchmark_mm.py:41, uncompiled), at S = 1024 / 2048 / 4096, each leg
    timed separately on the host cores torch was given.  Imported from oracle/ -- the checker, timed as a baseline only.
    `value` is the full path at 4096^3 (the size the GPU metric is quoted on)."""
    from oracle import torch_ref

    torch.manual_seed(2025 * 2024)
    legs = {}
    full = {}
    for S, share in ((1024, 0.1), (2048, 0.2), (4096, 0.7)):
        A, W = torch.randn(S, S).bfloat16(), torch.randn(S, S).bfloat16()
        A_i8, sa = torch_ref.quantize_int8(A, dim=1)
        W_i8, sw = torch_ref.quantize_int8(W, dim=1)

        def full_path():
            a, s0 = torch_ref.quantize_int8(A, dim=1)
            w, s1 = torch_ref.quantize_int8(W, dim=1)
            return torch_ref.scaled_mm_intmm(a, w.T, s0, s1.T)

        b = budget_s * share / 3
        tq, nq = _median_time(lambda: torch_ref.quantize_int8(A, dim=1), b)
        tm, nm = _median_time(lambda: torch._int_mm(A_i8, W_i8.T), b)
        tf, nf = _median_time(full_path, b)
        full[S] = tf
        legs[str(S)] = {"quantize_one_operand_ms": round(tq * 1e3, 3), "quantize_GB/s_algorithmic": round((3 * S * S + 2 * S) / tq * 1e-9, 2),
                        "int_mm_ms": round(tm * 1e3, 3), "int_mm_TOPS": round(2 * S ** 3 / tm * 1e-12, 3),
                        "full_path_ms": round(tf * 1e3, 3), "full_path_TOPS": round(2 * S ** 3 / tf * 1e-12, 3), "runs": [nq, nm, nf]}
    return {"value": legs["4096"]["full_path_TOPS"], "unit": "TOPS", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "configs[0] formula (2 x quantize_int8 in ATen + torch._int_mm + fp32 dequant scale -> bf16, the reference's fallback "
                      "benchmark_mm.py:41 fed by subclasses/int8.py:12-26) on randn bf16 A, W at S = 1024 / 2048 / 4096, seed 2025*2024, "
                      f"median of >= 3 runs per leg within a {budget_s:.0f} s budget; value = full path at 4096^3",
            "legs": legs}


def vendor_yardstick(cases, S):
    """Same-box library yardsticks at 4096^3 (event-timed, same method as detail_tops): torch._int_mm and torch._scaled_mm go to
    hipBLASLt, torch.mm (bf16) to hipBLASLt / rocBLAS.  They answer whether a vendor GEMM exceeds this repository's kernels on this chip
    under the same clocks; each entry is TOPS or the reason the library refused the layout."""
    out = {}

    def t(name, fn):
        try:
            fn()
            avg, _ = time_kernel(fn, iters=20)
            out[name] = round(2 * S ** 3 / (avg * 1e-3) * 1e-12, 1)
        except Exception as exc:
            out[name] = f"refused: {type(exc).__name__}: {str(exc)[:120]}"

    for lname in LAYOUTS:
        A, B, sA, sB = cases[("int8", lname)]
        t(f"torch._int_mm int8 {lname}", lambda A=A, B=B: torch._int_mm(A, B))
        A8, B8, _, _ = cases[("fp8", lname)]
        ra, cb = sA.float(), sB.float()
        t(f"torch._scaled_mm fp8 rowwise-scaled {lname}", lambda A8=A8, B8=B8: torch._scaled_mm(A8, B8, scale_a=ra, scale_b=cb, out_dtype=torch.bfloat16))
    one = torch.ones((), device=sA.device)
    A8, B8, _, _ = cases[("fp8", "A@B.T")]
    t("torch._scaled_mm fp8 tensorwise-scaled A@B.T", lambda: torch._scaled_mm(A8, B8, scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
    Ab = torch.randn(S, S, device=sA.device).bfloat16()
    Bb = torch.randn(S, S, device=sA.device).bfloat16()
    t("torch.mm bf16 A@B.T", lambda: torch.mm(Ab, Bb.T))
    t("torch.mm bf16 A@B", lambda: torch.mm(Ab, Bb))
    return out


def vendor_yardstick_small(dev):
    """the same-box library yardsticks at the sizes where fixed costs dominate -- 1024^3, 2048^3 and the five GEMM
    shapes of a Llama-2-470m step (8192 tokens; M x N x K) -- A row-major, B column-major (the layout hipBLASLt's int8 / fp8 paths take),
    every entry HIP-graph-replayed (no host launch path on either side).  TOPS per entry; `ours / vendor` for the int8 pair."""
    import gn_kernels

    out = {}
    shapes = [(1024, 1024, 1024), (2048, 2048, 2048), (8192, 1024, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (1024, 1024, 8192), (1024, 4096, 8192)]
    for M, N, K in shapes:
        A = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev)
        B = torch.randint(-128, 127, (N, K), dtype=torch.int8, device=dev).T
        A8, B8 = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn), torch.randn(N, K, device=dev).to(torch.float8_e4m3fn).T
        Ab, Bb = torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16().T
        sa, sb = torch.randn(M, 1, device=dev).div(128).bfloat16(), torch.randn(1, N, device=dev).div(128).bfloat16()
        ra, cb = sa.float(), sb.float()
        row = {}
        for name, fn in (("ours int8_mm -> int32", lambda: gn_kernels.int8_mm(A, B)),
                         ("vendor torch._int_mm -> int32", lambda: torch._int_mm(A, B)),
                         ("ours scaled_mm int8 -> bf16", lambda: gn_kernels.scaled_mm(A, B, sa, sb)),
                         ("ours scaled_mm fp8 -> bf16", lambda: gn_kernels.scaled_mm(A8, B8, sa, sb)),
                         ("vendor torch._scaled_mm fp8 rowwise -> bf16", lambda: torch._scaled_mm(A8, B8, scale_a=ra, scale_b=cb, out_dtype=torch.bfloat16)),
                         ("vendor torch.mm bf16", lambda: torch.mm(Ab, Bb))):
            try:
                med = time_kernel_graph(fn, per_graph=20, replays=5)
                row[name] = round(2 * M * N * K / (med * 1e-3) * 1e-12, 1)
            except Exception as exc:
                row[name] = f"refused: {type(exc).__name__}: {str(exc)[:80]}"
        a, b = row.get("ours int8_mm -> int32"), row.get("vendor torch._int_mm -> int32")
        if isinstance(a, float) and isinstance(b, float):
            row["ours / vendor (int8 -> int32)"] = round(a / b, 3)
        out[f"{M}x{N}x{K}"] = row
    return out


TRAIN_DETAIL_BUDGET_S = 420   # no new training run is started after this many seconds of detail_train


def train_detail(world, rank, quick=False):
    """The tok/s half of BASELINE.json's metric, outside the timed region: Llama INT8 mixed-precision training steps through
    train_bench.run (the harness restating llm_pretrain.py), torch.compile(get_loss) as the reference runs it, bf16 beside INT8-MP.
    At N > 1 this is DDP over RCCL with the reference's GLOBAL batch of 16 sequences (strong scaling, llm_pretrain.py:93,151)."""
    import gc

    import train_bench

    runs = [("llama-1b", ["--model", "1b", "--batch_size", "16"], 6, 2, ("mixed_precision", "none"))]
    if world == 1:
        runs.insert(0, ("llama-2-470m", ["--model", "470m", "--batch_size", "4"], 10, 3, ("mixed_precision", "none")))
    else:
        # beside the reference's fixed global batch (2 sequences per GPU at N = 8: the per-GPU GEMMs shrink with N), the weak-scaling
        # form: 16 sequences PER GPU, the per-GPU step of the N = 1 run plus the gradient all-reduce
        runs.append((f"llama-1b weak (global batch {16 * world})", ["--model", "1b", "--batch_size", str(16 * world)], 6, 2, ("mixed_precision",)))
    res = {}
    t_start = time.perf_counter()
    for name, margs, steps, warm, quants in runs:
        if quick:
            steps, warm = 2, 1
        for q in quants:
            key = f"{name} {'int8_mp' if q == 'mixed_precision' else 'bf16'} compiled"
            # time budget for the untimed detail (rank 0 decides for everybody, so that no rank enters a collective alone)
            late = torch.tensor([1.0 if time.perf_counter() - t_start > TRAIN_DETAIL_BUDGET_S else 0.0], device="cuda")
            if world > 1:
                dist.broadcast(late, src=0)
            if late.item() > 0:
                if rank == 0:
                    res[key] = f"skipped: detail_train had used its {TRAIN_DETAIL_BUDGET_S} s budget"
                continue
            try:
                torch._dynamo.reset()
                a = train_bench.parse_args(margs + ["--quantize", q, "--steps", str(steps), "--warmup", str(warm), "--compile"])
                r = train_bench.run(a)
                if rank == 0:
                    res[key] = {"tok/s": r["value"], "ms_per_step": r["ms_per_step"], "body_linear_TOPS": r["body_linear_TOPS"],
                                "max_mem_GiB": r.get("max_mem_GiB"), "loss_first_last": r["loss_first_last"], "workload": r["config"]["workload"]}
            except Exception as exc:   # detail only: never a reason to lose the bench line
                if rank == 0:
                    res[key] = f"failed: {type(exc).__name__}: {str(exc)[:200]}"
            gc.collect()
            torch.cuda.empty_cache()
        a, b = res.get(f"{name} int8_mp compiled"), res.get(f"{name} bf16 compiled")
        if isinstance(a, dict) and isinstance(b, dict):
            res[f"{name} int8_mp / bf16"] = round(a["tok/s"] / b["tok/s"], 3)
    return res


def bare_mfma_rate():
    """tools/probe_mfma_rate (built by __graft_entry__.build()): what a register-resident MFMA loop sustains on this chip right now."""
    import subprocess

    exe = ROOT / "tools" / "probe_mfma_rate"
    if not exe.exists():
        return None
    try:
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-detail", action="store_true", help="skip the untimed per-kernel sweep")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-train", action="store_true", help="skip the untimed Llama training tok/s runs (detail_train)")
    ap.add_argument("--quick-train", action="store_true", help="detail_train with 2 steps per run (smoke use)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    import gn_kernels
    from gn_kernels import _cabi

    _cabi.load().gnk_init()
    torch.manual_seed(2025 * 2024 + rank)
    S = 4096
    cases = make_inputs(S, dev)
    step_keys = [(d, l) for d in ("int8", "fp8") for l in LAYOUTS]

    def step():
        for k in step_keys:
            gn_kernels.scaled_mm(*cases[k])

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed, before the W warm-up steps: steady clocks (see prewarm); every rank does the same
    prewarm_calls, prewarm_ms = prewarm(step)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    ops_per_step = len(step_keys) * 2 * S ** 3
    value = world * ops_per_step * args.steps / elapsed * 1e-12

    out = {
        "metric": "scaled INT8/FP8 mm TOPS @4096 (A@B.T, A@B, A.T@B)",
        "value": round(value, 2),
        "unit": "TOPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int8+fp8_e4m3 (int32 / fp32 accumulate, bf16 out)",
        "data": "synthetic",
        "prewarm": {"steps": prewarm_calls, "ms": round(prewarm_ms, 1),
                    "why": "untimed, before the W warm-up steps: after an idle period the chip needs ~30 ms of load to reach the clocks it then holds "
                           "(tools/ramp_probe.py: step 375-397 us for the first 20 steps, 332 us from step 80 on); the timed region is steady state"},
        "config": {"workload": "configs[1]: scaled INT8 + scaled FP8 mm, M=N=K=4096, layouts A@B.T / A@B / A.T@B, 6 GEMMs per step",
                   "parallelism": f"replicas x{world}" if world > 1 else "single GPU"},
    }

    if rank == 0:
        # dominant kernel for the roofline: scaled INT8 A@B.T (the training forward layout)
        avg_ms, med_ms = time_kernel(lambda: gn_kernels.scaled_mm(*cases[("int8", "A@B.T")]), iters=100)
        achieved = 2 * S ** 3 / (avg_ms * 1e-3) * 1e-12
        out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_DENSE_8BIT_TOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / PEAK_DENSE_8BIT_TOPS, 4), "traffic": PMC_TRAFFIC_BYTES,
                           "kernel": "gemm_w4_kernel<256x256 tile, 4 waves (one per SIMD, 128x128 wave tiles), hand-scheduled asm loop, int8 v_mfma_i32_16x16x64_i8, A K-major, B K-major, row/col epilogue>",
                           "traffic_note": "memory-side bytes per launch measured in separate rocprofv3 --pmc passes over this kernel (profiles/r03x_pmc_int8_nt_4096_w4.txt; rounds 1-2, 8-wave kernels: r01e / r02d_pmc_int8_nt_4096.txt, within 0.2 %): "
                                           "FETCH_SIZE 50.6 MB x 2 (gfx950 correction) + WRITE_SIZE 33.6 MB (= the algorithmic 32 MiB of C); algorithmic bytes 67.1 MB -- "
                                           "the reads beyond 33.5 MB are per-XCD re-reads served by the Infinity Cache",
                           "avg_launch_us": round(avg_ms * 1e3, 2), "per_launch_event_pair_median_us": round(med_ms * 1e3, 2),
                           "frac_at_per_launch_event_pair_median": round(2 * S ** 3 / (med_ms * 1e-3) * 1e-12 / PEAK_DENSE_8BIT_TOPS, 4),
                           "timing_note": "avg_launch_us = one HIP event pair around 100 back-to-back launches / 100 (launch gaps included); the per-launch event-pair "
                                          "median (rounds 1-2's method, what rocprofv3's per-dispatch durations also track) carries ~3 us of event-record packets per "
                                          "launch: an empty kernel reads 6.0 us that way, 2.3-2.6 us back to back (tools/probe_launch.hip, DESIGN 7)",
                           "algorithmic_ops_per_launch": 2 * S ** 3}
        bare = bare_mfma_rate()
        if bare and "i8_16x16x64_tops" in bare:
            out["roofline"]["bare_mfma_loop_tops"] = bare["i8_16x16x64_tops"]
            out["roofline"]["frac_of_bare_mfma_loop"] = round(achieved / bare["i8_16x16x64_tops"], 4)
            out["roofline"]["bare_mfma_note"] = ("tools/probe_mfma_rate run on this GPU beside the bench: v_mfma_i32_16x16x64_i8 back to back from registers, 8 waves on "
                                                 "every CU, random operands -- the rate the matrix pipe sustains at the clock the chip holds under this load "
                                                 f"(fp8 16x16x128: {bare.get('f8_16x16x128_tflops')} TFLOP/s)")
        elif bare:
            out["roofline"]["bare_mfma_loop_tops"] = bare
        if not args.no_detail:
            detail = {}
            for (d, l), a in cases.items():
                avg, _ = time_kernel(lambda a=a: gn_kernels.scaled_mm(*a), iters=20)
                detail[f"{d} {l} {S}"] = round(2 * S ** 3 / (avg * 1e-3) * 1e-12, 1)
            graph = {}
            for s in (1024, 2048):
                small = make_inputs(s, dev)
                for d in ("int8", "fp8"):
                    for l in LAYOUTS:
                        avg, _ = time_kernel(lambda a=small[(d, l)]: gn_kernels.scaled_mm(*a), iters=20)
                        detail[f"{d} {l} {s}"] = round(2 * s ** 3 / (avg * 1e-3) * 1e-12, 1)
                        try:
                            med = time_kernel_graph(lambda a=small[(d, l)]: gn_kernels.scaled_mm(*a))
                            graph[f"{d} {l} {s}"] = round(2 * s ** 3 / (med * 1e-3) * 1e-12, 1)
                        except Exception as exc:   # graph capture is a measurement aid, never a reason to lose the bench line
                            graph[f"{d} {l} {s}"] = f"capture failed: {type(exc).__name__}"
            out["detail_tops"] = detail
            out["detail_tops_hipgraph_replay"] = graph
            # quantize pass: algorithmic bytes = 3*R*C + 2*R (read bf16 once, write int8 + bf16 scales)
            from subclasses import quantize_int8

            # event-timed through the op (Python dispatch + ctypes ~ 17-20 us per call is inside the number: at
            # 4096^2 the kernels themselves take 12 us row-wise / 19 us column-wise, profiles/*kernel_stats*)
            q = {}
            for R, C in ((S, S), (4 * S, 2 * S)):
                x = torch.randn(R, C, device=dev).bfloat16()
                for name, dim in (("rowwise (reduce contiguous dim)", 1), ("colwise (reduce strided dim)", 0)):
                    avg, _ = time_kernel(lambda dim=dim: quantize_int8(x, dim=dim), iters=30)
                    nbytes = 3 * R * C + 2 * (R if dim == 1 else C)
                    gbs = nbytes / (avg * 1e-3) * 1e-9
                    q[f"{R}x{C} {name}"] = {"GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK_HBM_GBS, 4), "us": round(avg * 1e3, 2)}
                    try:   # the same pass replayed from a HIP graph: kernel time without the Python launch path
                        med = time_kernel_graph(lambda dim=dim: quantize_int8(x, dim=dim), per_graph=10)
                        q[f"{R}x{C} {name}"].update({"hipgraph_us": round(med * 1e3, 2), "hipgraph_GB/s": round(nbytes / (med * 1e-3) * 1e-9, 1),
                                                     "hipgraph_frac_of_8TBs": round(nbytes / (med * 1e-3) * 1e-9 / PEAK_HBM_GBS, 4)})
                    except Exception as exc:
                        q[f"{R}x{C} {name}"]["hipgraph_us"] = f"capture failed: {type(exc).__name__}"
                del x
            out["detail_quantize_bf16"] = q
            # next rows: fused dual-direction quantize and the MX path, HIP-graph replay timing
            nxt = {}
            try:
                from gn_kernels import mx
                from gn_kernels import quantize as qmod

                x = torch.randn(32768, 2048, device=dev).bfloat16()        # Llama-1B activation shape
                _, _, cm = qmod.quantize_int8_rowwise_colmax(x)
                t_sep = time_kernel_graph(lambda: (quantize_int8(x, dim=1), quantize_int8(x, dim=0)), per_graph=5)
                t_fus = time_kernel_graph(lambda: (qmod.quantize_int8_rowwise_colmax(x), qmod.quantize_int8_colwise_given(x, cm)), per_graph=5)
                nxt["quantize both dims 32768x2048 bf16"] = {"separate_us": round(t_sep * 1e3, 1), "fused_pair_us": round(t_fus * 1e3, 1),
                                                            "fused_GB/s_at_6B_per_elem": round(6 * x.numel() / (t_fus * 1e-3) * 1e-9, 1)}
                t_q = time_kernel_graph(lambda: mx.quantize_mx(x, mx.FP4), per_graph=5)
                nxt["quantize_mx 32768x2048 bf16 -> fp4"] = {"us": round(t_q * 1e3, 1), "GB/s": round(x.numel() * (2 + 0.5 + 1 / 32) / (t_q * 1e-3) * 1e-9, 1)}
                del x
                A4 = torch.randint(0, 256, (S, S // 2), dtype=torch.uint8, device=dev).view(mx.FP4)
                B4 = torch.randint(0, 256, (S, S // 2), dtype=torch.uint8, device=dev).view(mx.FP4).T
                s4a = torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device=dev).view(mx.E8M0)
                s4b = torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device=dev).view(mx.E8M0)
                t_mm = time_kernel_graph(lambda: mx.mxfp4_mm(A4, B4, s4a, s4b), per_graph=10)
                nxt[f"mxfp4_mm {S}^3"] = {"us": round(t_mm * 1e3, 1), "TFLOP/s": round(2 * S ** 3 / (t_mm * 1e-3) * 1e-12, 1),
                                          "frac_of_fp4_dense_peak_10066": round(2 * S ** 3 / (t_mm * 1e-3) * 1e-12 / 10066.0, 4)}
                del A4, B4, s4a, s4b
                A8 = torch.randn(S, S, device=dev).to(torch.float8_e4m3fn)
                B8 = torch.randn(S, S, device=dev).to(torch.float8_e4m3fn).T
                s8a = torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device=dev).view(mx.E8M0)
                s8b = torch.randint(120, 134, (S, S // 32), dtype=torch.uint8, device=dev).view(mx.E8M0)
                t_mm8 = time_kernel_graph(lambda: mx.mxfp8_mm(A8, B8, s8a, s8b), per_graph=10)
                nxt[f"mxfp8_mm {S}^3"] = {"us": round(t_mm8 * 1e3, 1), "TFLOP/s": round(2 * S ** 3 / (t_mm8 * 1e-3) * 1e-12, 1),
                                          "frac_of_fp8_dense_peak_5033": round(2 * S ** 3 / (t_mm8 * 1e-3) * 1e-12 / PEAK_DENSE_8BIT_TOPS, 4)}
                del A8, B8, s8a, s8b
                from gn_kernels import bitnet as bop

                w = (torch.randn(2 * S, 2 * S, device=dev) * 0.02).bfloat16()      # BitNet weight path, 8192^2 bf16
                sc = bop.bitnet_absmean(w)
                t_am = time_kernel_graph(lambda: bop.bitnet_absmean(w), per_graph=10)
                t_qp = time_kernel_graph(lambda: bop.bitnet_quantize_weight(w, sc, packed=True), per_graph=10)
                nxt[f"bitnet {2 * S}x{2 * S} bf16"] = {"absmean_us": round(t_am * 1e3, 1), "absmean_GB/s": round(2 * w.numel() / (t_am * 1e-3) * 1e-9, 1),
                                                     "quantize+pack2bit_us": round(t_qp * 1e3, 1),
                                                     "quantize+pack2bit_GB/s": round(2.25 * w.numel() / (t_qp * 1e-3) * 1e-9, 1)}
                del w
            except Exception as exc:   # detail only: never a reason to lose the bench line
                nxt["error"] = f"{type(exc).__name__}: {exc}"
            out["detail_next_rows"] = nxt
            out["detail_vendor_yardstick_4096_tops"] = vendor_yardstick(cases, S)
            try:
                out["detail_vendor_yardstick_small_tops_hipgraph_replay"] = vendor_yardstick_small(dev)
            except Exception as exc:   # detail only
                out["detail_vendor_yardstick_small_tops_hipgraph_replay"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
    del cases
    torch.cuda.empty_cache()
    if not args.no_train:        # every rank takes part (DDP at N > 1); rank 0 reports
        tr = train_detail(world, rank, quick=args.quick_train)
        if rank == 0:
            out["detail_train"] = tr
    if rank == 0:
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
