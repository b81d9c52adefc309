"""A re-quantising forward (training mode / cache_operands off: what the reference does on every call, lora.py:142, :49-50) makes
the weight-side operands inside its activation pass (spq_fwd_args.prepare) instead of launching their preparation.  Same
arithmetic, so the output must be bit-identical to the separately prepared one and to the cached-operand forward -- on shapes
where the row work is spread over the 16-row activation kernel and on shapes that fall back to the preparation launch."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


def build(pkg, M, K, N, r, bits, qtype, per_channel=True, seed=0):
    from llm_qat_on_gpt2_amd import synthetic as S
    W, bias, A, B, x0, x1 = S.make_workload(M, K, N, max(r, 1), seed=seed, batch=1)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: max(r, 1), 32: 0}, {bits: qtype, 32: None},
                                 per_channel=per_channel)
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        if r > 0:
            layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    return layer, x0.to(DEV)


# (M, K, N, r, bits, qtype): fused = 16-row activation kernel and <= 8 weight rows per workgroup
CASES = [
    (8192, 768, 3072, 64, 4, "minmax"),      # headline: 6 rows per workgroup
    (4096, 768, 768, 64, 4, "minmax"),       # 3 rows per workgroup
    (2048, 768, 2304, 64, 8, "minmax"),      # 8-bit, 18 rows -> preparation launch
    (1024, 3072, 768, 64, 4, "minmax"),      # K = 3072: 12 rows -> preparation launch
    (4096, 256, 1000, 100, 4, "minmax"),     # N not a multiple of 128 (padding rows), rank 100 (two 64-wide blocks)
    (4096, 1024, 512, 0, 4, "minmax"),       # no LoRA branch
    (4096, 1024, 1024, 64, 6, "log"),        # log quantizers: two-limb activation operand (F16X3)
    (4096, 200, 256, 16, 4, "minmax"),       # K % 64 != 0: the generic activation kernel, preparation launch
    (20480, 128, 256, 16, 4, "minmax"),      # M >= 16384: the 32-row panel kernel, preparation launch
    (4096, 768, 3072, 64, 4, "minmax", False),   # per-tensor scales
]


# how the library serves spq_fwd_args.prepare: "role" (default) = extra workgroups of the streaming activation launch where that
# kernel runs; "launch" = always the ordinary preparation launch, issued by the library
MODES = {"role": {}, "launch": {"SPQ_PREP_ROLE": "0"}}


@pytest.fixture(params=list(MODES))
def prep_mode(request, pkg):
    import os
    old = os.environ.get("SPQ_PREP_ROLE")
    pkg._lib.set_switch("SPQ_PREP_ROLE", MODES[request.param].get("SPQ_PREP_ROLE"))
    yield request.param
    pkg._lib.set_switch("SPQ_PREP_ROLE", old)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_fused_prepare_is_bit_identical(pkg, case, prep_mode):
    M, K, N, r, bits, qtype = case[:6]
    layer, x = build(pkg, M, K, N, r, bits, qtype, per_channel=case[6] if len(case) > 6 else True)
    with torch.no_grad():
        y_cached = layer(x)                               # eval mode: operands prepared once, reused
        assert torch.equal(layer(x), y_cached)
        layer.cache_operands = False
        layer.fuse_prepare = False
        y_sep = layer(x)                                  # separate preparation launch on every call
        layer.fuse_prepare = True
        y_fused = layer(x)                                # row work inside the activation pass where the shape allows
        y_fused2 = layer(x)
    assert torch.equal(y_sep, y_cached)
    assert torch.equal(y_fused, y_cached), f"max abs diff {float((y_fused - y_cached).abs().max()):.3e}"
    assert torch.equal(y_fused2, y_cached)
    assert bool(torch.isfinite(y_fused).all())


def test_fused_prepare_tracks_weight_updates(pkg):
    """Training-mode semantics: a weight written between two forwards is picked up by the next one (no stale operands)."""
    layer, x = build(pkg, 4096, 768, 768, 64, 4, "minmax")
    layer.train()
    with torch.no_grad():
        y0 = layer(x)
        layer.linear.weight.data.mul_(1.5)                # through .data: no version bump (main_sp.py:81-99 writes this way)
        layer.lora_adapters["4bit"].lora_B.data.mul_(-1.0)
        pkg.calibrate_layer(layer, 4, [x])                # scales follow the new weights (train_sp.py:362-364 does this every step)
        layer.train()
        y1 = layer(x)
        layer.eval(); layer.invalidate_operand_cache()
        y1_eval = layer(x)
    assert not torch.equal(y0, y1)
    assert torch.equal(y1, y1_eval)
