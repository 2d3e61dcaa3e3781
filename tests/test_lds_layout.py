"""Host simulation of the GEMM's LDS staging / fragment-read index math (csrc/gnk_lds_layout.h)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_lds_layout_round_trip_and_bank_conflicts(tmp_path):
    exe = tmp_path / "lds_layout_sim"
    subprocess.check_call(["g++", "-O1", "-std=c++17", str(ROOT / "tests" / "lds_layout_sim.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
