"""CPU oracle of the BitNet weight path (numpy) -- TEST INFRASTRUCTURE ONLY, like the rest of ``oracle/``.

Restates, in plain numpy fp32 / integer arithmetic, the reference's
  get_bitnet_scale        subclasses/bitnet.py:110-112   mean(|f32(x)|)
  quantize_bitnet_weight  subclasses/bitnet.py:115-118   clip(round(f32(w) / max(scale, eps)), -1, 1) -> int8
  _pack_i2_in_i8          subclasses/bitnet.py:185-195
  _unpack_i2_in_i8        subclasses/bitnet.py:198-201
and the forward of _BitNetTrainingLinear (:123-144) on top of the int8 oracle (``oracle.quantize_int8`` / ``oracle.scaled_mm``).

Parity status: the four helpers are PINNED by tests/golden/g8_bitnet.npz (generated from the reference by
tools/gen_golden_bitnet.py); quantize / pack / unpack bit-exactly, the scale to 1e-6 relative (the mean is an fp32 sum whose
order is torch-internal: this restatement sums in float64 and rounds once).  The Linear forward / backward golden comes from
the reference's autograd function with its one drifted call repaired (the 0-dim scale viewed as (1, 1)).
"""
from __future__ import annotations

import numpy as np


def get_bitnet_scale(x_f32: np.ndarray) -> np.float32:
    """x_f32: the tensor already widened to fp32 (bf16 / fp16 -> fp32 is exact)."""
    return np.float32(np.abs(x_f32.astype(np.float64)).sum() / x_f32.size)


def quantize_bitnet_weight(w_f32: np.ndarray, scale: np.float32, eps: float = 1e-5) -> np.ndarray:
    div = np.maximum(np.float32(scale), np.float32(eps))
    v = np.rint(w_f32.astype(np.float32) / div)            # fp32 IEEE division, round half to even
    return np.clip(v, -1.0, 1.0).astype(np.int8)


def pack_i2(x: np.ndarray) -> np.ndarray:
    """(M, N) int8 -> (M, N / 4) int8; int8 wrap-around on the left shift, like torch."""
    u = x.view(np.uint8).astype(np.uint32)
    out = ((u[:, 0::4] << 6) & 0xFF) | ((u[:, 1::4] & 3) << 4) | ((u[:, 2::4] & 3) << 2) | (u[:, 3::4] & 3)
    return out.astype(np.uint8).view(np.int8)


def unpack_i2(x: np.ndarray) -> np.ndarray:
    """(M, N) packed -> (M, 4 N) int8: each 2-bit field sign-extended."""
    u = x.view(np.uint8).astype(np.int32)
    fields = np.stack([(u >> 6) & 3, (u >> 4) & 3, (u >> 2) & 3, u & 3], axis=-1)
    fields = np.where(fields >= 2, fields - 4, fields)
    return fields.reshape(x.shape[0], -1).astype(np.int8)
