"""CPU: the C-ABI shared library loads and exports every symbol include/spq.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    import llm_qat_on_gpt2_amd as pkg
    if not os.path.exists(pkg._lib.LIB_PATH):
        g.build()
    return pkg._lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "spq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    import llm_qat_on_gpt2_amd as pkg
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/spq.h but not exported by libspq.so"
        assert n in pkg._lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(pkg._lib.SIGNATURES) == names


def test_version_and_error_string(lib):
    assert lib.spq_version() == 101
    assert isinstance(lib.spq_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    """Pure host-side argument checks: rejected before any HIP call."""
    import llm_qat_on_gpt2_amd as pkg
    rc = lib.spq_finish_scale(None, None, 4, 8, 0, 1, 1e-5, None, None, None)
    assert rc == -1 and b"null pointer" in lib.spq_last_error()
    rc = lib.spq_fakequant(None, 1, 1, 1, None, None, 0, 8, 0, 1, None, None, 0, None)
    assert rc == -1
    assert lib.spq_fwd_workspace_bytes(0, 1, 1, 0, pkg._lib.PATH_F32) == 0
    assert lib.spq_fwd_workspace_bytes(128, 64, 64, 8, pkg._lib.PATH_F32) >= 128 * 64 * 4 + 128 * 8 * 4
    assert lib.spq_stats_workspace_bytes(8192, 768, 1, 1) > 0
    with pytest.raises(pkg._lib.SpqError):
        pkg._lib.check(rc, "spq_fakequant")


def test_struct_layout_matches_header():
    import llm_qat_on_gpt2_amd as pkg
    # 4 int64 + 6 int + 4 ptr + 5 ptr + float(+pad) + 2 ptr + size_t + 2 ptr + t_out + 3 int(+pad) + a_limb_scale
    assert ctypes.sizeof(pkg._lib.FwdArgs) == 4 * 8 + 6 * 4 + 9 * 8 + 8 + 3 * 8 + 2 * 8 + 8 + 16 + 8
