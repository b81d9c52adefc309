"""Host logic of the contraction's split-K form, no GPU: which shapes the library would split, what SPQ_SPLIT_K forces,
and that the workspace it asks for covers the split it would take.  (The kernel itself: tests/test_gpu_splitk.py.)"""
import os

import pytest


@pytest.fixture()
def lib():
    import llm_qat_on_gpt2_amd as p
    old = os.environ.get("SPQ_SPLIT_K")
    yield p._lib
    p._lib.set_switch("SPQ_SPLIT_K", old)


F16X2, F16X3, I8 = 2, 4, 5


def test_automatic_choice(lib):
    lib.set_switch("SPQ_SPLIT_K", None)
    s = lib.load().spq_debug_split_k
    assert s(8192, 3072, 768, 64, F16X2) == 2          # mlp c_proj at 8 x 1024 tokens: 384 tiles of 50 stages on 768 slots
    assert s(8192, 768, 768, 64, F16X2) == 1           # attn c_proj: 14 stages, too short to pay
    assert s(8192, 768, 3072, 64, F16X2) == 1          # c_fc: 1536 tiles fill the chip
    assert s(32768, 3072, 768, 64, F16X2) == 1         # config 3: 1536 tiles
    assert s(8192, 4096, 1024, 64, F16X3) == 1         # 512 tiles x 3 would need two rounds: measured slower
    assert s(4096, 3072, 768, 64, F16X2) in (2, 3, 4)  # 192 tiles
    assert s(8192, 3072, 770, 64, F16X2) == 1          # ragged N: the scalar-store instantiation has no split form


def test_forced_and_off(lib):
    s = lib.load().spq_debug_split_k
    lib.set_switch("SPQ_SPLIT_K", "0")
    assert s(8192, 3072, 768, 64, F16X2) == 1
    lib.set_switch("SPQ_SPLIT_K", "3")
    assert s(8192, 3072, 768, 64, F16X2) == 3          # 1152 units: two rounds, legal when forced
    assert s(8192, 768, 768, 64, F16X2) == 3           # 14 stages / 3 = 5 >= 4
    assert s(8192, 256, 768, 64, F16X2) == 1           # 6 stages / 3 < 4: not legal
    assert s(8192, 768, 3072, 64, F16X2) == 1          # tiles alone fill the slots


def test_workspace_covers_the_split(lib):
    w = lib.load().spq_fwd_workspace_bytes
    lib.set_switch("SPQ_SPLIT_K", "0")
    base = w(8192, 3072, 768, 64, F16X2)
    lib.set_switch("SPQ_SPLIT_K", None)
    auto = w(8192, 3072, 768, 64, F16X2)
    assert auto - base == 384 * 2 * 65536 + 8192       # one 64-KB register image per unit + the counters
    assert w(8192, 768, 3072, 64, F16X2) == w(8192, 768, 3072, 64, I8)      # no split-K scratch where nothing splits
    lib.set_switch("SPQ_SPLIT_K", "4")
    assert w(8192, 3072, 768, 64, F16X2) - base == 384 * 4 * 65536 + 8192
    assert w(8192, 3072, 768, 64, I8) == base          # the int8 kernels have no split form
