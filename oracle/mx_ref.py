"""CPU oracle of the MX path (numpy) -- TEST INFRASTRUCTURE ONLY, like the rest of ``oracle/``.

Restates, in plain numpy integer / fp32 arithmetic, the reference's
  quantize_mx          kernels/gn_kernels/utils.py:83-107  (scale rules :30-41 "nv", :45-49 "ocp"; FP4 table :52-80)
  dequantize_mxfp4     kernels/gn_kernels/utils.py:110-142
  pack_block_scales_nv kernels/gn_kernels/utils.py:21-26
and defines the expectation for mxfp4_mm (kernels/gn_kernels/cutlass_mm.py:80-86 -> csrc/cutlass_sm120a_fp4.cu:42-136,
which has no CPU implementation): the fp32 product of the dequantized operands, plus bias, cast to bf16.

Parity status: quantize_mx / dequantize_mxfp4 / pack_block_scales_nv PINNED by tests/golden/g7_quantize_mx.npz
(generated from the reference by tools/gen_golden_mx.py).  mxfp4_mm: parity unpinned (no reference output exists in
this container: the reference op is CUTLASS-on-Blackwell only); products of FP4 values and power-of-two scales are
exact in fp32, so the only freedom is the accumulation order -- tests state the tolerance.
"""
from __future__ import annotations

import numpy as np

DTYPE_AMAX = {"fp4": 6.0, "e4m3": 448.0, "e5m2": 57344.0}
POW2_LOG = {"fp4": 2, "e4m3": 8, "e5m2": 15}
FP4_LUT = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], dtype=np.float32)


def _fp8_code(y: np.ndarray, ebits: int, mbits: int) -> np.ndarray:
    """fp32 (already within the finite range) -> fp8 code, round to nearest even."""
    bias = (1 << (ebits - 1)) - 1
    u = y.view(np.uint32)
    sign = (u >> 31).astype(np.uint32) << 7
    mag = u & 0x7FFFFFFF
    e = (mag >> 23).astype(np.int64) - 127
    m = (mag & 0x7FFFFF).astype(np.int64)
    drop = 23 - mbits
    q = ((e + bias) << mbits) | (m >> drop)
    rem = m & ((1 << drop) - 1)
    half = 1 << (drop - 1)
    q = q + ((rem > half) | ((rem == half) & ((q & 1) == 1)))
    # subnormal targets: k * 2^(1 - bias - mbits), k by half-to-even
    k = np.rint(np.abs(y).astype(np.float64) * 2.0 ** (bias - 1 + mbits)).astype(np.int64)
    code = np.where(e >= 1 - bias, q, k)
    code = np.where(mag == 0, 0, code)
    return (sign | code.astype(np.uint32)).astype(np.uint8)


def quantize_mx(x: np.ndarray, fmt: str, method: str = "ocp"):
    """x: (M, N) float32 (upcast the input first) -> (uint8 data, uint8 E8M0 scales (M, N / 32))."""
    assert x.dtype == np.float32 and x.ndim == 2 and x.shape[1] % 32 == 0
    M, N = x.shape
    blocks = x.reshape(M, N // 32, 32)
    amax = np.abs(blocks).max(axis=-1).astype(np.float32)
    if method == "ocp":
        e = ((amax.view(np.uint32) >> 23) & 0xFF).astype(np.int64)
        bits = np.maximum(e - POW2_LOG[fmt], 0)
    elif method == "nv":
        sb = (amax / np.float32(DTYPE_AMAX[fmt])).astype(np.float32).view(np.uint32)
        e = (sb >> 23).astype(np.int64)
        m = (sb & 0x7FFFFF).astype(np.int64)
        up = ((e > 0) & (e < 254) & (m > 0)) | ((e == 0) & (m > 0x400000))
        bits = np.where(up, e + 1, e)
    else:
        raise ValueError(method)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        div = np.maximum((bits.astype(np.uint32) << 23).view(np.float32), np.float32(1e-12))
        y = (blocks / div[..., None]).astype(np.float32)
    y = np.clip(y, -np.float32(DTYPE_AMAX[fmt]), np.float32(DTYPE_AMAX[fmt])).astype(np.float32)
    scales = (bits & 0xFF).astype(np.uint8)
    if fmt == "fp4":
        a = np.abs(y)
        code = np.where(a <= 5.0, 6, 7)
        for bound, strict, c in ((3.5, True, 5), (2.5, False, 4), (1.75, True, 3), (1.25, False, 2), (0.75, True, 1), (0.25, False, 0)):
            code = np.where(a < bound if strict else a <= bound, c, code)
        code = (code | ((y.view(np.uint32) >> 31) << 3)).astype(np.uint8).reshape(M, N)
        q = (code[:, 0::2] | (code[:, 1::2] << 4)).astype(np.uint8)
    else:
        q = _fp8_code(y.reshape(M, N), *((4, 3) if fmt == "e4m3" else (5, 2)))
    return q, scales


def dequantize_mxfp4(q: np.ndarray, scales: np.ndarray) -> np.ndarray:
    M, N = q.shape[0], q.shape[1] * 2
    codes = np.stack([q & 0xF, q >> 4], axis=-1).reshape(M, N)
    s = (scales.astype(np.uint32) << 23).view(np.float32)
    return (FP4_LUT[codes].reshape(M, N // 32, 32) * s.reshape(M, N // 32, 1)).reshape(M, N).astype(np.float32)


def pack_block_scales_nv(scales: np.ndarray) -> np.ndarray:
    M, N = scales.shape
    out = scales.reshape(M // 128, 128, N // 4, 4).transpose(0, 2, 1, 3)
    out = out.reshape(-1, 4, 32, 4).transpose(0, 2, 1, 3).reshape(-1, 32, 16)
    return out.reshape(-1)


def mxfp4_mm(A: np.ndarray, Bt: np.ndarray, sA: np.ndarray, sB: np.ndarray, bias: np.ndarray | None = None) -> np.ndarray:
    """A (M, K/2), Bt = B.T (N, K/2) packed codes; scales (M, K/32), (N, K/32) -> fp32 (M, N) BEFORE the output cast,
    accumulated in float64 (every product is exact, so this is the exact value up to one final rounding)."""
    a = dequantize_mxfp4(A, sA).astype(np.float64)
    b = dequantize_mxfp4(Bt, sB).astype(np.float64)
    out = a @ b.T
    if bias is not None:
        out = out + bias.astype(np.float64).reshape(1, -1)
    return out


def _fp8_value(codes: np.ndarray, fmt: str) -> np.ndarray:
    """OCP e4m3fn / e5m2 byte codes -> float64 values (exact)."""
    c = codes.astype(np.int64)
    sign = np.where(c & 0x80, -1.0, 1.0)
    if fmt == "e4m3":
        e, m = (c >> 3) & 15, c & 7
        v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e - 7.0))
        v = np.where((e == 15) & (m == 7), np.nan, v)
    else:
        e, m = (c >> 2) & 31, c & 3
        v = np.where(e == 0, m / 4.0 * 2.0 ** -14, (1 + m / 4.0) * 2.0 ** (e - 15.0))
        v = np.where(e == 31, np.where(m == 0, np.inf, np.nan), v)
    return sign * v


def dequantize_mxfp8(q: np.ndarray, scales: np.ndarray, fmt: str) -> np.ndarray:
    """(rows, K) fp8 codes + (rows, K / 32) E8M0 bytes -> float64 values: code value * 2^(scale - 127)."""
    s = 2.0 ** (scales.astype(np.float64) - 127.0)
    return _fp8_value(q, fmt) * np.repeat(s, 32, axis=1)


def mxfp8_mm(A: np.ndarray, Bt: np.ndarray, sA: np.ndarray, sB: np.ndarray, fmt_a: str, fmt_b: str, bias: np.ndarray | None = None) -> np.ndarray:
    """MXFP8 product BEFORE the output cast, in float64 (include/gnk.h: gnk_mxfp8_mm).  A (M, K), Bt = B.T (N, K) byte codes."""
    out = dequantize_mxfp8(A, sA, fmt_a) @ dequantize_mxfp8(Bt, sB, fmt_b).T
    if bias is not None:
        out = out + bias.astype(np.float64).reshape(1, -1)
    return out
