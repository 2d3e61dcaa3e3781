"""The seven hot operators have two host paths above the same C ABI: the C++ registration (libgnk_torch_ops.so, the default) and the Python
+ ctypes `device_impl` bodies (used when the shim is not built, or with GNK_NO_TORCH_SHIM=1).  The rest of the GPU suite runs on the first;
this file runs the golden and oracle checks of those operators once more in a fresh child process on the second, and
checks from inside that process that the shim really stayed out."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

# reference goldens g1 (quantize_int8), g2 / g3 (int8_mm, scaled_mm), g4 (tile_scaled_mm), g6 (the Linear's three GEMMs + scaled_mm_t), and the
# oracle comparisons of the fused quantize pair: every operator in _cabi._SHIM_OPS
SELECTION = [
    "tests/test_gpu_gemm.py::test_int8_mm_golden_bit_exact",
    "tests/test_gpu_gemm.py::test_scaled_mm_int8_golden_bit_exact",
    "tests/test_gpu_gemm.py::test_tile_scaled_int8_deepseek_golden_bit_exact",
    "tests/test_gpu_gemm.py::test_tile_scaled_int8_jetfire_golden_bit_exact",
    "tests/test_gpu_quantize.py::test_golden_from_reference",
    "tests/test_gpu_quantize.py::test_rowwise_colmax_and_colwise_given_equal_the_separate_passes",
    "tests/test_gpu_subclasses.py::test_mixed_precision_linear_golden_bit_exact",
]


def test_python_ctypes_host_path_passes_the_goldens():
    env = dict(os.environ, GNK_NO_TORCH_SHIM="1", GNK_EXPECT_NO_SHIM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "120", "--timeout-method", "thread",
                        *SELECTION, "tests/test_gpu_host_paths.py::test_child_sees_no_shim"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail


def test_child_sees_no_shim():
    """Runs in both processes: in the child (GNK_EXPECT_NO_SHIM=1) the C++ registration must be absent, in the parent present."""
    sys.path[:0] = [str(ROOT / "quantized-training_amd")]
    import gn_kernels  # noqa: F401
    from gn_kernels import _cabi

    assert _cabi.torch_shim_loaded() == (os.environ.get("GNK_EXPECT_NO_SHIM") != "1")
