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
