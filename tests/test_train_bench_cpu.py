"""The training harness (train_bench.py) must keep running: single process and 2-rank DDP over gloo on the CPU with the
bf16 (unquantized) model -- the INT8 ops have no CPU implementation by design, so the quantized path is a GPU test."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--model", "tiny", "--quantize", "none", "--device", "cpu", "--seq_len", "32", "--steps", "2", "--warmup", "1"]


def _last_json(out: str) -> dict:
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_single_process_step_runs_and_reports_tokens_per_second():
    r = subprocess.run([sys.executable, str(ROOT / "train_bench.py"), *ARGS, "--batch_size", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["unit"] == "tok/s" and out["value"] > 0 and out["n_gpus"] == 1
    assert abs(out["value"] - 2 * 32 * 2 / (out["ms_per_step"] * 2 / 1e3)) / out["value"] < 0.02   # tok/s = batch * seq * steps / dt


def test_two_rank_ddp_over_gloo_splits_the_global_batch():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           str(ROOT / "train_bench.py"), *ARGS, "--batch_size", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and "DDP x2" in out["config"]["workload"] and out["value"] > 0


def test_two_rank_ddp_int8_mixed_precision_compiled_over_gloo():
    """configs[4]'s code path, rehearsed on the CPU: DDP(model) over gloo + torch.compile(get_loss) + MixedPrecisionLinearWeight
    parameters (llm_pretrain.py:126-132,184).  The ops' CPU backend is the oracle (tests/_train_bench_oracle_worker.py, test-only);
    both ranks must finish, agree on the averaged gradients' effect (same loss trajectory as the eager run) and report tok/s."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    args = ["--model", "tiny", "--quantize", "mixed_precision", "--device", "cpu", "--seq_len", "32", "--steps", "2", "--warmup", "1", "--batch_size", "4"]
    outs = {}
    for tag, port, extra in (("eager", "29641", []), ("compiled", "29642", ["--compile", "--compile_backend", "aot_eager"])):
        r = subprocess.run(base + ["--master-port", port, str(ROOT / "tests" / "_train_bench_oracle_worker.py"), *args, *extra],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[tag] = _last_json(r.stdout)
    assert outs["compiled"]["n_gpus"] == 2 and "torch.compile" in outs["compiled"]["config"]["workload"]
    assert outs["compiled"]["loss_first_last"] == outs["eager"]["loss_first_last"]      # same ops on the same data: identical losses
