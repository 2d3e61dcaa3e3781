"""The drop-in boundary without a GPU: the C-ABI library loads and exports every symbol declared in
include/gnk.h, the gn_kernels op schemas / Meta impls / wrapper checks match the reference, and there
is NO CPU implementation in the product (the ops fail loudly on CPU tensors, as in the reference)."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

import gn_kernels
from gn_kernels import _cabi

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "gnk.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ["gnk_quantize_int8", "gnk_int8_mm", "gnk_scaled_mm", "gnk_tile_scaled_mm", "gnk_fp8_mm", "gnk_last_error",
              "gnk_abi_version", "gnk_init", "gnk_scaled_mm_ws", "gnk_scaled_mm_workspace_bytes"]:
        assert s in syms


def test_library_loads_and_exports_every_declared_symbol():
    assert _cabi.library_path().exists(), "build libgnk_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    L = ctypes.CDLL(str(_cabi.library_path()))
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} is declared in include/gnk.h but not exported"
    assert _cabi.load().gnk_abi_version() == _cabi.ABI_VERSION


def test_cabi_rejects_bad_arguments_without_touching_a_gpu():
    L = _cabi.load()
    rc = L.gnk_scaled_mm(None, _cabi.I8, None, _cabi.F32, None, _cabi.BF16, None, None, _cabi.BF16, 0, 0, 4, 4, 4, 4, 1, 1, 4, None)
    assert rc == -1 and b"int8 or both fp8" in L.gnk_last_error()
    rc = L.gnk_quantize_int8(None, _cabi.I8, 4, 4, 4, 1, None, 4, 1, None, 1e-12, 0, 0, 0, None, 0, None)
    assert rc == -1 and b"not bf16/f16/f32" in L.gnk_last_error()
    rc = L.gnk_int8_mm(None, None, None, -1, 4, 4, 4, 1, 4, 1, None)
    assert rc == -1
    assert L.gnk_int8_mm(None, None, None, 0, 4, 4, 4, 1, 4, 1, None) == 0      # empty problem: nothing to do
    assert L.gnk_quantize_int8_workspace_bytes(128, 64, 1, 128) == 2 * 512        # strided reduction over 64 rows: 2 partial rows of R floats
    assert L.gnk_quantize_int8_workspace_bytes(128, 8192, 1, 128) == 32 * 512     # never more than 32 partial rows
    assert L.gnk_quantize_int8_workspace_bytes(128, 64, 64, 1) == 0


def test_public_names_match_the_reference_package():
    ref_all = ["int4_mm", "scaled_int4_mm", "scaled_fp8_mm", "fp8_mm", "mxfp4_mm", "nvfp4_mm", "_triton_conv2d", "int8_conv2d",
               "scaled_int8_conv2d", "_triton_mm", "int8_mm", "scaled_mm", "quantize_mx", "quantize_nvfp4", "quantize_nvfp4_triton",
               "dequantize_mxfp4", "pack_block_scales_nv"]
    for name in ref_all:
        assert hasattr(gn_kernels, name), name
    with pytest.raises(NotImplementedError):
        gn_kernels.int4_mm(torch.zeros(2, 2, dtype=torch.int8), torch.zeros(2, 2, dtype=torch.int8))
    with pytest.raises(NotImplementedError):
        gn_kernels.nvfp4_mm(None, None, None, None, None)
    assert str(torch.ops.gn_kernels.mxfp4_mm.default._schema) == "gn_kernels::mxfp4_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B, Tensor? bias) -> Tensor"


def test_op_schemas_match_the_reference():
    ops = torch.ops.gn_kernels
    assert str(ops.int8_mm.default._schema) == "gn_kernels::int8_mm(Tensor A, Tensor B) -> Tensor"
    assert str(ops.scaled_mm.default._schema) == "gn_kernels::scaled_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor"
    assert str(ops.tile_scaled_mm.default._schema) == "gn_kernels::tile_scaled_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor"
    assert str(ops.fp8_mm.default._schema) == "gn_kernels::fp8_mm(Tensor A, Tensor B) -> Tensor"
    assert str(ops.scaled_fp8_mm.default._schema) == "gn_kernels::scaled_fp8_mm(Tensor A, Tensor B, Tensor row_scale, Tensor col_scale) -> Tensor"
    assert str(ops.cutlass_fp8_mm.default._schema) == "gn_kernels::cutlass_fp8_mm(Tensor A, Tensor B) -> Tensor"
    assert str(ops.cutlass_scaled_fp8_mm.default._schema) == "gn_kernels::cutlass_scaled_fp8_mm(Tensor A, Tensor B, Tensor scale_A, Tensor scale_B) -> Tensor"


def test_meta_impls_give_reference_shapes_and_dtypes():
    A = torch.empty(48, 64, dtype=torch.int8, device="meta")
    B = torch.empty(64, 32, dtype=torch.int8, device="meta")
    assert gn_kernels.int8_mm(A, B).dtype is torch.int32 and gn_kernels.int8_mm(A, B).shape == (48, 32)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        out = gn_kernels.scaled_mm(A, B, torch.empty(48, 1, dtype=dt, device="meta"), torch.empty(1, 32, dtype=dt, device="meta"))
        assert out.dtype is dt and out.shape == (48, 32)
        out = gn_kernels.scaled_mm(A, B, torch.empty(48, 1, dtype=dt, device="meta"), torch.empty(1, 1, dtype=dt, device="meta"))
        assert out.shape == (48, 32)
    out = gn_kernels.scaled_mm(A, B, torch.empty(3, 2, dtype=torch.bfloat16, device="meta"), torch.empty(2, 4, dtype=torch.bfloat16, device="meta"))
    assert out.dtype is torch.bfloat16 and out.shape == (48, 32)
    F8 = torch.empty(48, 64, dtype=torch.float8_e4m3fn, device="meta")
    G8 = torch.empty(32, 64, dtype=torch.float8_e4m3fn, device="meta").T
    assert gn_kernels.fp8_mm(F8, G8).dtype is torch.bfloat16
    out = gn_kernels.scaled_fp8_mm(F8, G8, torch.empty(48, 1, device="meta"), torch.empty(1, 32, device="meta"))
    assert out.dtype is torch.bfloat16 and out.shape == (48, 32)
    x = torch.empty(5, 7, dtype=torch.bfloat16, device="meta")
    q, s = gn_kernels.quantize_int8(x.T, dim=1)
    assert q.dtype is torch.int8 and q.shape == (7, 5) and q.stride() == x.T.stride() and s.shape == (7, 1) and s.dtype is torch.bfloat16


def test_wrapper_asserts_follow_the_reference():
    i8 = lambda *s: torch.zeros(*s, dtype=torch.int8, device="meta")  # noqa: E731
    bf = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device="meta")  # noqa: E731
    with pytest.raises(AssertionError):      # dtype class mismatch
        gn_kernels.scaled_mm(i8(4, 8), torch.zeros(8, 4, dtype=torch.float8_e4m3fn, device="meta"), bf(4, 1), bf(1, 4))
    with pytest.raises(AssertionError):      # scale dtype mismatch
        gn_kernels.scaled_mm(i8(4, 8), i8(8, 4), bf(4, 1), torch.zeros(1, 4, device="meta"))
    with pytest.raises(AssertionError):      # inner dims
        gn_kernels.scaled_mm(i8(4, 8), i8(7, 4), bf(4, 1), bf(1, 4))
    with pytest.raises(AssertionError):      # (N, 1) column scale is routed to tile scaling and fails its check
        gn_kernels.scaled_mm(i8(4, 8), i8(8, 4), bf(4, 1), bf(4, 1))
    with pytest.raises(AssertionError):      # non-contiguous row scale
        gn_kernels.scaled_mm(i8(4, 8), i8(8, 4), bf(4, 2)[:, :1], bf(1, 4))
    with pytest.raises(AssertionError):
        gn_kernels.int8_mm(bf(4, 8), i8(8, 4))
    with pytest.raises(AssertionError):      # fp8_mm wants row-major x column-major
        gn_kernels.fp8_mm(torch.zeros(4, 8, dtype=torch.float8_e4m3fn, device="meta"), torch.zeros(8, 4, dtype=torch.float8_e4m3fn, device="meta"))
    # mixing the two fp8 formats is allowed
    gn_kernels.scaled_mm(torch.zeros(4, 8, dtype=torch.float8_e4m3fn, device="meta"), torch.zeros(8, 4, dtype=torch.float8_e5m2, device="meta"), bf(4, 1), bf(1, 4))


def test_product_has_no_cpu_path():
    """Like the reference (Meta + device impls only): a CPU call must fail, never silently compute."""
    if torch._C._dispatch_has_kernel_for_dispatch_key("gn_kernels::scaled_mm", "CPU"):
        pytest.skip("the oracle_cpu_ops test fixture is already active in this process")
    A = torch.zeros(4, 8, dtype=torch.int8)
    with pytest.raises(NotImplementedError):
        gn_kernels.scaled_mm(A, A.T.contiguous(), torch.ones(4, 1), torch.ones(1, 4))


def test_product_sources_never_touch_the_oracle():
    for path in (ROOT / "quantized-training_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".h", ".cpp") and "build" not in path.parts:
            text = path.read_text()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), path        # Python imports
            assert not re.search(r"#\s*include\s+[\"<][^\n]*oracle", text), path                # C/C++ includes
            assert "libgnk_oracle" not in text and "CDLL(" not in text.replace("ctypes.CDLL(str(_LIB_PATH))", ""), path


def test_triton_mm_with_float_operands_is_a_plain_library_gemm():
    """reference triton_mm.py:131-143 / benchmark_mm.py:110: `_triton_mm(A_f16, B_f16, acc_dtype=torch.float16)`, an unchecked timing row.
    Here: torch.mm with fp32 accumulation, cast to out_dtype (default A's dtype)."""
    g = torch.Generator().manual_seed(0)
    A, B = torch.randn(32, 64, generator=g).bfloat16(), torch.randn(64, 48, generator=g).bfloat16()
    out = gn_kernels._triton_mm(A, B, acc_dtype=torch.float16)
    assert out.dtype is torch.bfloat16 and out.shape == (32, 48)
    torch.testing.assert_close(out.float(), A.float() @ B.float(), atol=0.25, rtol=2e-2)
    assert gn_kernels._triton_mm(A.float(), B.float(), out_dtype=torch.bfloat16).dtype is torch.bfloat16
    with pytest.raises(NotImplementedError):
        gn_kernels._triton_mm(A, B.float())
