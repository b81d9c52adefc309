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


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of both argument structs as gcc lays include/spq.h out, against the ctypes mirrors."""
    import subprocess
    import llm_qat_on_gpt2_amd as pkg
    structs = {"spq_fwd_args": pkg._lib.FwdArgs, "spq_prepare_args": pkg._lib.PrepareArgs}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "spq.h")}"', 'int main(void) {']
    for cname, st in structs.items():
        src.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            src.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", str(c), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, st in structs.items():
        assert int(out[cname]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"

