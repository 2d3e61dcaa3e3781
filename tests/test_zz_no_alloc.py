"""The entry point that allocates nothing: gnk_quantize_int8_noalloc (include/gnk.h) against gnk_quantize_int8, through the C ABI.

It has no counterpart in the reference; the bar is bit-equality with the existing entry point (which the other tests pin to the
reference goldens and the oracle).  Two calls in a row share the library's built-in scratch buffer, the second on a smaller input:
stale column maxima of the first call must not leak into it.  Named test_zz_* so that pytest collects it last."""
import pytest
import torch

from gn_kernels import _cabi

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _call(L, fn_name, x, dim, sr=False, seed=0, offset=0, workspace=True):
    """quantize a 2-D device tensor along `dim` through one of the two C entry points; returns (q, scale)."""
    d = dim % 2
    R, C = (x.shape[0], x.shape[1]) if d == 1 else (x.shape[1], x.shape[0])
    sxr, sxc = (x.stride(0), x.stride(1)) if d == 1 else (x.stride(1), x.stride(0))
    q = torch.empty_strided(x.shape, x.stride(), dtype=torch.int8, device=x.device)
    sqr, sqc = (q.stride(0), q.stride(1)) if d == 1 else (q.stride(1), q.stride(0))
    scale = torch.empty(R, dtype=x.dtype, device=x.device)
    st = _cabi.stream_of(x)
    code = _cabi.DTYPE_CODE[x.dtype]
    if fn_name == "gnk_quantize_int8":
        nbytes = L.gnk_quantize_int8_workspace_bytes(R, C, sxr, sxc)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
        rc = L.gnk_quantize_int8(x.data_ptr(), code, R, C, sxr, sxc, q.data_ptr(), sqr, sqc, scale.data_ptr(), 1e-12, int(sr), seed, offset,
                                 ws.data_ptr(), nbytes, st)
    else:
        rc = L.gnk_quantize_int8_noalloc(x.data_ptr(), code, R, C, sxr, sxc, q.data_ptr(), sqr, sqc, scale.data_ptr(), 1e-12, int(sr), seed,
                                         offset, st)
    return rc, q, scale


def _same(a, b):
    view = {1: torch.int8, 2: torch.int16, 4: torch.int32}[a.element_size()]
    return torch.equal(a.contiguous().view(view), b.contiguous().view(view))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_no_alloc_matches_existing_entry_point_on_two_calls_in_a_row(dtype):
    L = _cabi.load()
    assert L.gnk_init() == 0
    torch.manual_seed(11)
    # first call: large values in every column, so every scratch word ends up large
    big = (torch.randn(2048, 4096, device=DEV) * 50).to(dtype)
    # second call: SMALLER input with small values (and a zero column): a leaked maximum of call 1 would change its scales
    small = (torch.randn(384, 1024, device=DEV) * 0.01).to(dtype)
    small[:, 5] = 0
    torch.cuda.synchronize()
    for x in (big, small):
        for dim in (0, 1):                    # 0 = strided reduction (uses the scratch buffer), 1 = contiguous reduction (does not)
            free_before = torch.cuda.mem_get_info()[0]
            rc, q, s = _call(L, "gnk_quantize_int8_noalloc", x, dim)
            assert rc == 0, L.gnk_last_error()
            rc0, q0, s0 = _call(L, "gnk_quantize_int8", x, dim)
            assert rc0 == 0, L.gnk_last_error()
            torch.cuda.synchronize()
            assert _same(q, q0), f"int8 data differs ({tuple(x.shape)}, dim {dim})"
            assert _same(s, s0), f"scales differ ({tuple(x.shape)}, dim {dim})"
            assert torch.cuda.mem_get_info()[0] >= free_before - (64 << 20)   # nothing grabbed behind torch's back (torch's own pools may grow)
    # transposed-view spelling and stochastic rounding share the same path
    xt = small.T
    rc, q, s = _call(L, "gnk_quantize_int8_noalloc", xt, 1, sr=True, seed=1234, offset=1 << 40)
    rc0, q0, s0 = _call(L, "gnk_quantize_int8", xt, 1, sr=True, seed=1234, offset=1 << 40)
    torch.cuda.synchronize()
    assert rc == 0 and rc0 == 0 and _same(q, q0) and _same(s, s0)


def test_no_alloc_rejects_sizes_beyond_its_scratch_buffer():
    L = _cabi.load()
    max_rows = L.gnk_quantize_int8_noalloc_max_rows()
    assert max_rows == 65536
    x = torch.zeros(8, max_rows + 8, device=DEV, dtype=torch.bfloat16)          # strided reduction over max_rows + 8 logical rows
    rc, _, _ = _call(L, "gnk_quantize_int8_noalloc", x, 0)
    assert rc == -2 and b"at most 65536 rows" in L.gnk_last_error()
    ok = torch.randn(8, max_rows, device=DEV, dtype=torch.bfloat16)             # exactly the maximum is accepted
    rc, q, s = _call(L, "gnk_quantize_int8_noalloc", ok, 0)
    rc0, q0, s0 = _call(L, "gnk_quantize_int8", ok, 0)
    torch.cuda.synchronize()
    assert rc == 0 and rc0 == 0 and _same(q, q0) and _same(s, s0)
    rc, _, _ = _call(L, "gnk_quantize_int8_noalloc", x, 1)                      # contiguous reduction needs no scratch: any size
    assert rc == 0
