"""The streaming activation pass (xpass_stream_kernel: three-slot LDS-DMA ring per 64-column chunk, levels by reciprocal
multiply with an IEEE-division fallback for lanes on a rounding tie) against the panel kernels it replaces (SPQ_XPASS_STREAM=0)
and against the oracle.  Integer levels are an exact result: one wrong level moves y by about scale * weight, thousands of times
the 1e-5 bound, so "bit-identical y" and "y within the bound of the oracle" both pin the level pass."""
import os

import numpy as np
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
def _restore_env(pkg):
    old = os.environ.get("SPQ_XPASS_STREAM")
    yield
    pkg._lib.set_switch("SPQ_XPASS_STREAM", old)


def build(pkg, M, K, N, r, bits, per_channel=True, seed=0):
    from llm_qat_on_gpt2_amd import synthetic as S
    W, bias, A, B, x0, x1 = S.make_workload(M, K, N, max(r, 1), seed=seed, batch=1)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: max(r, 1), 32: 0}, {bits: "minmax", 32: None},
                                 per_channel=per_channel)
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    return layer, x0.to(DEV), (W, bias, A, B)


def _set(name, value):
    import llm_qat_on_gpt2_amd as p
    p._lib.set_switch(name, value)


def run_modes(layer, x, modes=("0", "16", "32", "1")):
    out = {}
    with torch.no_grad():
        for m in modes:
            _set("SPQ_XPASS_STREAM", m)
            out[m] = layer(x).clone()
    return out


# (M, K, N, r, bits, per_channel, operand path)
CASES = [
    (8192, 768, 256, 64, 4, True, "auto"),        # headline rows: 32-row streaming workgroups on every CU
    (4100, 768, 128, 64, 4, True, "auto"),        # ragged M (the last workgroup's rows clamp), 16-row workgroups
    (512, 3072, 128, 64, 4, True, "auto"),        # K = 3072: 48 chunks through the ring
    (16432, 256, 128, 16, 8, True, "auto"),       # M >= 16384 (was the 32-row panel kernel), rank 16, 8-bit
    (2048, 64, 128, 64, 4, True, "auto"),         # one chunk: no steady state
    (2048, 128, 128, 64, 6, True, "auto"),        # two chunks
    (4096, 768, 256, 64, 8, False, "auto"),       # per-tensor input scale: int8 levels (SPQ_PATH_I8)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_streaming_pass_matches_the_panel_kernels(pkg, case):
    M, K, N, r, bits, pc, path = case
    layer, x, _ = build(pkg, M, K, N, r, bits, per_channel=pc, seed=K + bits)
    out = run_modes(layer, x)
    assert bool(torch.isfinite(out["0"]).all())
    # the panel kernel sums the LoRA-down product in another order (8 k-slices per chunk against 4); the level pass is exact either way
    for m in ("16", "32", "1"):
        assert_close_y(out[m], out["0"], f"mode {m}", 1e-5)
    assert torch.equal(out["16"], out["32"]) and torch.equal(out["1"], out["32"])


def _tie_rows(scale, bits, M, gen):
    """Rows whose quotients x / scale sit on, one ulp below and one ulp above the rounding ties k + 1/2 (and on the integers)."""
    K = scale.numel()
    qmax = 2 ** (bits - 1) - 1
    j = torch.randint(-qmax - 2, qmax + 2, (M, K), generator=gen).float()
    half = torch.randint(0, 2, (M, K), generator=gen).float() * 0.5
    x = ((j + half) * scale[None, :]).float()
    nudge = torch.randint(-1, 2, (M, K), generator=gen).to(torch.int32)
    x = torch.where(x != 0, (x.view(torch.int32) + nudge).view(torch.float32), x)          # +-1 ulp of the product
    assert bool(torch.isfinite(x).all())
    return x


@pytest.mark.parametrize("bits,pc", [(4, True), (8, True), (8, False)])
def test_levels_on_rounding_ties(pkg, bits, pc):
    from oracle import ref_cpu as O
    M, K, N, r = 1024, 256, 128, 16
    layer, _, (W, bias, A, B) = build(pkg, M, K, N, r, bits, per_channel=pc, seed=11)
    key = f"{bits}bit"
    qx = layer.quantizers_input[key]
    gen = torch.Generator().manual_seed(5)
    scale = qx.qparams_for(K)[0].detach().reshape(-1).cpu()
    if scale.numel() == 1:
        scale = scale.expand(K).clone()
    assert scale.numel() == K
    x = _tie_rows(scale, bits, M, gen)
    lv_ref = torch.clamp(torch.round(x / scale[None, :]), -(2 ** (bits - 1) - 1), 2 ** (bits - 1) - 1)   # IEEE quotient, ties to even
    assert 0.1 < float(((x / scale[None, :]) % 1 == 0.5).float().mean()) < 0.6                          # the ties are really there
    out = run_modes(layer, x.to(DEV), modes=("0", "32", "16"))
    assert torch.equal(out["32"], out["16"])
    assert_close_y(out["32"], out["0"], "stream vs panel", 1e-6)      # (the LoRA-down sum in another order; one wrong level would be ~1e-2)
    # and against the checker: same levels -> y within the bound
    lora = layer.lora_adapters[key]

    def q(src, cd):
        s = O.QuantState(bits, "minmax", cd, src.scale.numel() > 1)
        s.scale, s.zero_point, s.calibrated = src.scale.detach().cpu(), src.zero_point.detach().cpu(), True
        return s
    oracle = O.OracleLayer(W, bias, A, B, q(qx, -1), q(layer.quantizers_weight[key], 0), q(lora.quantize_A, 1), q(lora.quantize_B, 1),
                           float(lora.scaling), bits)
    assert torch.equal(oracle.qx.levels(x).reshape(M, K), lv_ref)
    assert_close_y(out["32"].cpu(), oracle.forward(x), "ties", 1e-5)


def test_scales_outside_the_reciprocal_range(pkg):
    """Columns whose scale is huge (reciprocal subnormal), subnormal or zero leave the reciprocal path for the division; whatever
    the division gives (level 0, clamped) is what the panel kernels give.  (NaN / Inf inputs: tests/test_gpu_nan.py.)"""
    M, K, N, r, bits = 512, 128, 128, 16, 4
    layer, x, _ = build(pkg, M, K, N, r, bits, seed=3)
    qx = layer.quantizers_input[f"{bits}bit"]
    with torch.no_grad():
        s = qx.scale.reshape(-1)
        s[3] = 2.0 ** 110; s[7] = 2.0 ** 127; s[12] = 2.0 ** -140; s[20] = 0.0; s[50] = 2.0 ** -120
        qx._epoch += 1
        x = x.reshape(M, K).clone()
        x[:, 7] = 3.0e38; x[5, 12] = 1e-41; x[6, 50] = 2.0 ** -121 * 1.5
    layer.invalidate_operand_cache()
    out = run_modes(layer, x, modes=("0", "32", "16"))
    for m in ("32", "16"):
        assert torch.equal(torch.isnan(out[m]), torch.isnan(out["0"]))
        assert torch.equal(torch.nan_to_num(out[m], nan=0.0), torch.nan_to_num(out["0"], nan=0.0))


# ---- the limb form (A8 = 3): any other input quantizer on the streaming kernel, against the panel kernels it replaces
def build_q(pkg, M, K, N, r, bits, qtype, per_channel=True, symmetric=True, seed=0):
    from llm_qat_on_gpt2_amd import synthetic as S
    W, bias, A, B, x0, x1 = S.make_workload(M, K, N, max(r, 1), seed=seed, batch=1)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: max(r, 1), 32: 0}, {bits: qtype, 32: None}, per_channel=per_channel)
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(bits)
    if not symmetric:
        for q in (layer.quantizers_input[key], layer.quantizers_weight[key], layer.lora_adapters[key].quantize_A, layer.lora_adapters[key].quantize_B):
            q.symmetric = False
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    return layer, x0.to(DEV)


LIMB_CASES = [
    # M,    K,    N,   r,  bits, qtype,   per_channel, symmetric
    (8192, 1024, 256, 64, 6, "log", True, True),          # config 5's layers: 32-row workgroups
    (4100, 768, 128, 64, 6, "log", True, True),           # ragged M, 16-row workgroups
    (512, 4096, 128, 64, 4, "log", True, True),           # 64 chunks through the ring
    (2048, 64, 128, 16, 8, "log", False, True),           # one chunk, per-tensor range
    (2048, 128, 128, 64, 10, "log", True, True),          # > 8 bits: no level table
    (4096, 256, 128, 32, 4, "minmax", True, False),       # asymmetric min-max
    (4096, 256, 128, 32, 16, "minmax", True, True),       # 16-bit min-max: limbs
    (16432, 256, 128, 16, 5, "log", True, False),         # asymmetric log, M >= 16384
]


@pytest.mark.parametrize("case", LIMB_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_limb_form_matches_the_panel_kernels(pkg, case):
    M, K, N, r, bits, qtype, pc, sym = case
    layer, x = build_q(pkg, M, K, N, r, bits, qtype, pc, sym)
    old = os.environ.get("SPQ_XPASS_STREAM_LIMBS")
    try:
        with torch.no_grad():
            pkg._lib.set_switch("SPQ_XPASS_STREAM_LIMBS", "0")
            y_panel = layer(x).clone()
            assert layer._last_path == pkg._lib.PATH_F16X3
            pkg._lib.set_switch("SPQ_XPASS_STREAM_LIMBS", None)
            y_stream = layer(x).clone()
            layer.cache_operands = False                   # + the weight rows as extra workgroups of the same launch
            y_role = layer(x).clone()
    finally:
        pkg._lib.set_switch("SPQ_XPASS_STREAM_LIMBS", old)
    assert_close_y(y_stream, y_panel, "limb stream vs panel", 1e-6)   # (the panel kernel sums the LoRA-down product in another order)
    assert torch.equal(y_role, y_stream)
