"""No-GPU checks of the C-ABI library: it loads, and exports every function include/mudpt.h declares."""
import ctypes as C
import os

import pytest

from mudpt_amd import capi, build


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        build.build_library()
    return capi.load()


def test_header_functions_are_exported_and_bound(lib):
    declared = capi.declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mudpt.h but not exported"
        assert name in capi.SIGNATURES, f"{name} has no ctypes signature in mudpt_amd/capi.py"
    assert sorted(capi.SIGNATURES) == declared


def test_abi_version(lib):
    assert lib.mudpt_abi_version() == capi.ABI_VERSION


def test_padded_len_is_host_only(lib):
    assert [lib.mudpt_attention_padded_len(L) for L in (1, 32, 33, 77, 201, 224)] == [32, 32, 64, 96, 224, 224]


def test_argument_errors_do_not_touch_the_gpu(lib):
    assert lib.mudpt_create(None, None) == 1
    assert b"null" in lib.mudpt_last_error()
    cfg = capi.Config(224, 16, 768, 12, 12, 512, 12, 8, 77, 512, 4, 0, 11, 4, 0)  # DEEP_PROMPT_DEPTH 0
    h = C.c_void_p()
    assert lib.mudpt_create(C.byref(cfg), C.byref(h)) == 1
    assert b"PROMPT_DEPTH should be > 0" in lib.mudpt_last_error()  # trainers/mudpt.py:52
    cfg = capi.Config(224, 16, 768, 12, 12, 512, 12, 8, 77, 512, 4, 12, 11, 4, 7)  # unknown dtype
    assert lib.mudpt_create(C.byref(cfg), C.byref(h)) == 1
    with pytest.raises(AssertionError):
        capi.check(1, "create")


def test_allreduce_entry_point_validates_before_touching_rccl(lib):
    assert lib.mudpt_allreduce_grads(None, None, None) == 1 and b"null model" in lib.mudpt_last_error()


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "mudpt_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle: product must not depend on it"


def test_host_e4m3_conversion_matches_torch():
    """mudpt_e4m3_from_f32 (the conversion mudpt_set_weight applies to the frozen weights for the e4m3 second pass of the parity mode's
    vision tower) is OCP e4m3fn with round-to-nearest-even and saturation at +-448: bit for bit torch's float8_e4m3fn on in-range values."""
    import ctypes as C
    import torch
    from mudpt_amd import capi
    lib = capi.load()
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(20000, generator=g) * s for s in (1e-3, 0.02, 0.5, 4.0, 100.0)] +
                  [torch.tensor([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 3 * 2.0 ** -10, 0.0625 + 2.0 ** -8, 17.0, 18.0, 19.0, 463.9, 240.0, 232.0])])
    for shift in (0, 3, -2):
        out = torch.zeros(x.numel(), dtype=torch.uint8)
        assert lib.mudpt_e4m3_from_f32(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), shift) == 0
        ref = (x * 2.0 ** shift).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        keep = ~((ref & 0x7f) == 0) | (out & 0x7f == 0)  # +-0: the sign of a value that rounds to zero is kept either way
        assert torch.equal(out[keep], ref[keep]), (out != ref).nonzero()[:5]
        assert (((out & 0x7f) == 0) == ((ref & 0x7f) == 0)).all()
    big = torch.tensor([464.0, 480.0, 1e6, -1e6, float("inf")])
    out = torch.zeros(5, dtype=torch.uint8)
    lib.mudpt_e4m3_from_f32(C.c_void_p(big.data_ptr()), C.c_void_p(out.data_ptr()), 5, 0)
    assert out.tolist() == [0x7e, 0x7e, 0x7e, 0xfe, 0x7e]  # saturates, never the NaN code
