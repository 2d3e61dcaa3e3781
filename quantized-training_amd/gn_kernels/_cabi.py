"""ctypes binding of libgnk_hip.so (include/gnk.h) -- the only bridge between Python and the
hand-written gfx950 kernels.  There is no CPU or eager fallback here: if the library is missing,
or a call fails, the op raises."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

# GNK_LIBRARY: tuning only (an A/B build of the same sources, e.g. `make ABLATE=1 OUT=...`); the shipped library is the default
_LIB_PATH = Path(os.environ.get("GNK_LIBRARY") or Path(__file__).resolve().parent / "libgnk_hip.so")
ABI_VERSION = 2   # include/gnk.h GNK_ABI_VERSION

BF16, F16, F32, I8, E4M3, E5M2, I32 = range(7)
DTYPE_CODE = {
    torch.bfloat16: BF16,
    torch.float16: F16,
    torch.float32: F32,
    torch.int8: I8,
    torch.float8_e4m3fn: E4M3,
    torch.float8_e5m2: E5M2,
    torch.int32: I32,
}

_lib = None

# ---- C++ registration of the hot operators (csrc/torch_shim/gnk_torch_ops.cpp) ------------------------------------------------------
# Same kernels, same C ABI; only the per-call host path changes (dispatcher -> C++ -> C ABI instead of dispatcher -> Python -> ctypes).
# GNK_NO_TORCH_SHIM=1 keeps the Python implementations (A/B timing of the host path, and the tests' second leg).
_SHIM_PATH = Path(__file__).resolve().parent / "libgnk_torch_ops.so"
_SHIM_OPS = ("int8_mm", "scaled_mm", "scaled_mm_t", "tile_scaled_mm", "quantize_int8", "quantize_int8_rowwise_colmax", "quantize_int8_colwise_given")
_shim_loaded = None


def torch_shim_loaded() -> bool:
    """Load libgnk_torch_ops.so once (it registers the device implementations of _SHIM_OPS); False when switched off or not built."""
    global _shim_loaded
    if _shim_loaded is None:
        _shim_loaded = False
        if os.environ.get("GNK_NO_TORCH_SHIM", "0") in ("", "0") and not os.environ.get("GNK_LIBRARY"):
            if _SHIM_PATH.exists() and _LIB_PATH.exists():
                torch.ops.load_library(str(_SHIM_PATH))
                _shim_loaded = True
            else:
                import warnings

                warnings.warn(f"{_SHIM_PATH.name} is not built (make -C quantized-training_amd/csrc): gn_kernels ops take the slower Python "
                              "host path (same kernels)")
    return _shim_loaded


def device_impl(lib, name: str):
    """Decorator: register fn as the device ("CUDA" key) implementation of gn_kernels::name unless the C++ shim already provides it."""
    def deco(fn):
        if name in _SHIM_OPS and torch_shim_loaded():
            return fn
        return torch.library.impl(lib, name, "CUDA")(fn)
    return deco


def library_path() -> Path:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load libgnk_hip.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build it with `make -C quantized-training_amd/csrc -j8` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). gn_kernels has no fallback path."
        )
    L = ctypes.CDLL(str(_LIB_PATH))
    i64, vp, ci, u64, f32 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_float
    L.gnk_abi_version.restype = ci
    L.gnk_last_error.restype = ctypes.c_char_p
    L.gnk_init.restype = ci
    L.gnk_set_gemm_variant.argtypes = [ci]
    L.gnk_set_tile_rescale.argtypes = [ci]
    L.gnk_quantize_int8.argtypes = [vp, ci, i64, i64, i64, i64, vp, i64, i64, vp, f32, ci, u64, u64, vp, i64, vp]
    L.gnk_quantize_int8_workspace_bytes.argtypes = [i64, i64, i64, i64]
    L.gnk_quantize_int8_workspace_bytes.restype = i64
    L.gnk_quantize_int8_noalloc.argtypes = [vp, ci, i64, i64, i64, i64, vp, i64, i64, vp, f32, ci, u64, u64, vp]
    L.gnk_quantize_int8_noalloc_max_rows.restype = i64
    L.gnk_quantize_int8_rowwise_colmax_workspace_bytes.argtypes = [ci, i64, i64]
    L.gnk_quantize_int8_rowwise_colmax_workspace_bytes.restype = i64
    L.gnk_quantize_int8_rowwise_colmax.argtypes = [vp, ci, i64, i64, i64, vp, i64, vp, vp, f32, ci, u64, u64, vp, i64, vp]
    L.gnk_quantize_int8_colwise_given_absmax.argtypes = [vp, ci, i64, i64, i64, vp, i64, vp, vp, f32, ci, u64, u64, vp]
    L.gnk_quantize_mx.argtypes = [vp, ci, i64, i64, i64, vp, ci, i64, vp, i64, ci, vp]
    L.gnk_mxfp4_mm.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci] + [i64] * 7 + [vp]
    L.gnk_mxfp8_mm.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp, ci] + [i64] * 7 + [vp]
    L.gnk_bitnet_absmean_workspace_bytes.restype = i64
    L.gnk_bitnet_absmean.argtypes = [vp, ci, i64, vp, vp, i64, vp]
    L.gnk_bitnet_quantize_weight.argtypes = [vp, ci, i64, vp, f32, vp, ci, vp]
    L.gnk_bitnet_pack_i2.argtypes = [vp, i64, vp, vp]
    L.gnk_bitnet_unpack_i2.argtypes = [vp, i64, vp, vp]
    L.gnk_int8_mm.argtypes = [vp, vp, vp] + [i64] * 7 + [vp]
    L.gnk_scaled_mm.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci] + [i64] * 7 + [vp]
    L.gnk_scaled_mm_ws.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, ci, ci, ci] + [i64] * 7 + [vp, i64, vp]
    L.gnk_scaled_mm_workspace_bytes.argtypes = [ci, ci, i64, i64, i64]
    L.gnk_scaled_mm_workspace_bytes.restype = i64
    L.gnk_tile_scaled_mm.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, ci] + [i64] * 14 + [vp]
    L.gnk_fp8_mm.argtypes = [vp, ci, vp, ci, vp, ci] + [i64] * 7 + [vp]
    if L.gnk_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libgnk_hip.so ABI {L.gnk_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gnk_last_error().decode(errors="replace")
        raise RuntimeError(f"gn_kernels::{what} failed (status {rc}): {msg}")


def stream_of(t: torch.Tensor) -> int:
    """hipStream_t of the calling thread's current stream on t's device."""
    return torch._C._cuda_getCurrentRawStream(t.device.index)   # 0.1 us; torch.cuda.current_stream() builds a Stream object (2 us)


class device_of:
    """Make t's device current for the duration of a launch (no-op when it already is)."""

    def __init__(self, t: torch.Tensor):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if self.idx is not None and self.idx != cur:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False
