"""Split-K of the contraction (gemm_f16x2_t128_kernel<.., SK = true>): when the 128 x 128 tiles alone leave workgroup slots empty
(N <= 1024 at M = 8192: SURVEY 8(d) attn c_proj / mlp c_proj), S workgroups share a tile's k range and the last one to arrive sums
the S partial accumulators in FIXED order.  Checked here: the oracle bound at every S, run-to-run bit identity (the arrival
order must not matter), agreement with the unsplit kernel up to fp32 re-association, and that the automatic choice splits the
shapes it is meant for."""
import pytest
import torch

from helpers import assert_close_y

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


@pytest.fixture(autouse=True)
def _restore(pkg):
    import os
    old = os.environ.get("SPQ_SPLIT_K")
    yield
    pkg._lib.set_switch("SPQ_SPLIT_K", old)


def build(pkg, M, K, N, r, bits, qtype, seed=0):
    from oracle import ref_cpu as O
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, max(r, 1), seed=seed, batch=1)
    ol = O.build_calibrated_layer(W, bias, A, B, [x0, x1], bits, qtype, True, 64, max(r, 1)) if r else None
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: qtype, 32: None})
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        if r:
            layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    return layer, x0, ol


# (M, K, N, r, bits, qtype)
CASES = [
    (1024, 768, 256, 64, 4, "minmax"),       # 16 tiles, T = 14 stages: S = 2, 3
    (1024, 3072, 256, 64, 4, "minmax"),      # T = 50: S = 2, 3, 4
    (1000, 1024, 200, 16, 8, "minmax"),      # ragged M and N, rank 16
    (2048, 1024, 128, 0, 4, "minmax"),       # no LoRA stages
    (1024, 1024, 256, 64, 6, "log"),         # the three-product path (AL = 2): stage pairs stay together
    (8192, 768, 768, 64, 4, "minmax"),       # config 4 attn c_proj: 384 tiles on 768 slots
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_split_k_parity_and_determinism(pkg, case):
    M, K, N, r, bits, qtype = case
    layer, x0, ol = build(pkg, M, K, N, r, bits, qtype, seed=K + N)
    x = x0.to(DEV)
    outs = {}
    with torch.no_grad():
        for S in ("0", "2", "3", "4"):
            pkg._lib.set_switch("SPQ_SPLIT_K", S)
            y = layer(x).clone()
            for _ in range(3):                                         # the arrival order of a tile's units varies run to run
                assert torch.equal(layer(x), y), f"S={S}: not bit-identical run to run"
            outs[S] = y
    assert layer._last_path in (pkg._lib.PATH_F16X2, pkg._lib.PATH_F16X3)
    if ol is not None:
        ref = ol.forward(x0)
        for S, y in outs.items():
            assert_close_y(y, ref, f"S={S} vs oracle", 1e-5)
    for S in ("2", "3", "4"):
        assert_close_y(outs[S], outs["0"], f"S={S} vs unsplit", 5e-6)         # same products, fp32 sums re-associated
    # the split really happened where the library allows it (else the comparison above says nothing)
    assert any(not torch.equal(outs[S], outs["0"]) for S in ("2", "3", "4")), "no split variant differs from the unsplit kernel in any bit"


def test_auto_choice_splits_the_narrow_layers(pkg):
    """Default switches at M = 8192: mlp c_proj (K = 3072, N = 768: 384 tiles of 50 stages) is split -- the result differs from
    SPQ_SPLIT_K=0 in the last bits --; attn c_proj (14 stages: too short to pay) and c_fc (N = 3072 fills the chip) are not."""
    pkg._lib.set_switch("SPQ_SPLIT_K", None)
    for K, N, expect_split in ((3072, 768, True), (768, 768, False), (768, 3072, False)):
        layer, x0, _ = build(pkg, 8192, K, N, 64, 4, "minmax", seed=1)
        x = x0.to(DEV)
        with torch.no_grad():
            pkg._lib.set_switch("SPQ_SPLIT_K", None)
            y_auto = layer(x).clone()
            pkg._lib.set_switch("SPQ_SPLIT_K", "0")
            y_off = layer(x).clone()
        assert (not torch.equal(y_auto, y_off)) == expect_split, (K, N, expect_split)
        assert_close_y(y_auto, y_off, f"K={K} N={N}", 5e-6)
