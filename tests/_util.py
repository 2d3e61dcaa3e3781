"""Shared helpers for the parity tests: golden loading and layout construction."""
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"

LAYOUTS = {"nt": (False, True), "nn": (False, False), "tn": (True, False), "tt": (True, True)}
FLOAT_DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
FP8_DTYPES = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}


def load(name: str):
    return np.load(GOLDEN / name, allow_pickle=False)


def t_from_bits(arr: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """Rebuild a tensor stored by tools/gen_golden.py (bf16/fp16 as int16 bits, fp8 as uint8)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dtype in (torch.bfloat16, torch.float16):
        return t.view(dtype)
    if dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(dtype)
    return t.to(dtype)


def to_layout(x: torch.Tensor, column_major: bool) -> torch.Tensor:
    """benchmark_mm.py:33-34"""
    return x.T.contiguous().T if column_major else x.contiguous()


def bits_equal(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Bitwise equality that treats bf16/fp16/fp32 as raw bits (so -0.0 != 0.0 and NaN == NaN)."""
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    a, b = a.contiguous().cpu(), b.contiguous().cpu()
    view = {2: torch.int16, 4: torch.int32, 1: torch.int8}[a.element_size()]
    return torch.equal(a.view(view), b.view(view))
