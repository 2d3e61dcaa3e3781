import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT / "quantized-training_amd"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never silently pass without a device."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_cpu_ops():
    """Register the CPU ORACLE as the CPU backend of the gn_kernels ops -- in the TEST process only.
    The product registers Meta + device impls and nothing else (like the reference); this fixture is
    what lets the tensor-subclass host logic and the gloo DDP path run without a GPU."""
    import torch

    import gn_kernels  # noqa: F401  (defines the schemas)
    import oracle

    impl = torch.library.Library("gn_kernels", "IMPL")

    def scaled_mm_cpu(A, B, sa, sb):
        return oracle.scaled_mm(A, B, sa, sb)

    def int8_mm_cpu(A, B):
        return oracle.int8_mm(A, B)

    def _sr(sr, rng):
        return dict(seed=int(rng.item()), offset=0) if sr else {}

    def quantize_cpu(x, dim, sr, eps, rng=None):
        return oracle.quantize_int8(x, sr, dim=dim, eps=eps, **_sr(sr, rng))

    def rowwise_colmax_cpu(x, sr, eps, rng=None):
        q, s = oracle.quantize_int8(x, sr, dim=1, eps=eps, **_sr(sr, rng))
        return q, s, x.float().abs().amax(0)

    def colwise_given_cpu(x, absmax, sr, eps, rng=None):
        assert torch.equal(absmax, x.float().abs().amax(0)), "colwise_given: the absmax handed over is not the column absmax of x"
        return oracle.quantize_int8(x, sr, dim=0, eps=eps, **_sr(sr, rng))

    impl.impl("scaled_mm", scaled_mm_cpu, "CPU")
    impl.impl("scaled_mm_t", lambda A, B, sa, sb: oracle.scaled_mm(A, B, sa, sb).T.contiguous(), "CPU")   # its definition: the transpose, materialised
    impl.impl("tile_scaled_mm", scaled_mm_cpu, "CPU")
    impl.impl("int8_mm", int8_mm_cpu, "CPU")
    impl.impl("quantize_int8", quantize_cpu, "CPU")
    impl.impl("quantize_int8_rowwise_colmax", rowwise_colmax_cpu, "CPU")
    impl.impl("quantize_int8_colwise_given", colwise_given_cpu, "CPU")
    yield impl
