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
