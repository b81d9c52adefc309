"""GPU parity of the block-level pieces: SwitchableLayerNorm as one HIP pass and SPMLP with the GELU fused
into c_fc's store, against the fixtures the reference produced (tests/golden/blk_*.npz)."""
import types

import pytest
import torch

from helpers import cosim_block, rows_outside, assert_close_y
from test_blocks_cpu import load_blk

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


@pytest.mark.parametrize("name", ["ln_768", "ln_1024", "ln_64"])
def test_layernorm_kernel(pkg, name):
    """Summation order of mean / variance differs from ATen's CPU reduction: agreement to a few ulp of the row's scale."""
    meta, t = load_blk(name)
    ln = pkg.SwitchableLayerNorm(meta["C"], precision_levels=[4, 8, 32], eps=1e-5).to(DEV)
    x = t["x"].to(DEV)
    for p in (4, 8, 32):
        with torch.no_grad():
            ln.weights[str(p)].copy_(t[f"w_{p}"]); ln.biases[str(p)].copy_(t[f"b_{p}"])
            ln.set_precision(p)
            y = ln(x)
        ref = t[f"y_{p}"]
        d = (y.cpu().double() - ref.double()).abs()
        assert bool((d <= 2e-6 * ref.double().abs() + 2e-6).all()), f"{name}[{p}]: max abs diff {float(d.max()):.2e}"
        with torch.no_grad():
            assert torch.equal(y, ln(x))
    # autograd takes the composed formula and agrees with the kernel
    xg = x.clone().requires_grad_(True)
    yg = ln(xg)
    assert yg.grad_fn is not None and torch.allclose(yg.detach(), y, rtol=1e-5, atol=1e-5)


def build_mlp(pkg, meta, t):
    bits, r = meta["bits"], meta["r"]
    cfg = types.SimpleNamespace(n_embd=meta["E"], bit_widths=[bits, 32], lora_rank_per_bit={bits: r, 32: 0},
                                lora_alpha_per_bit={bits: meta["alpha"], 32: 0}, quantizer_per_bit={bits: meta["qtype"], 32: None},
                                per_channel_quantization=True)
    m = pkg.SPMLP(cfg, bit_widths=[bits, 32])
    key = f"{bits}bit"
    with torch.no_grad():
        for lin, s in ((m.c_fc, "f"), (m.c_proj, "p")):
            lin.linear.weight.copy_(t[f"W{s}"]); lin.linear.bias.copy_(t[f"b{s}"])
            lin.lora_adapters[key].lora_A.copy_(t[f"A{s}"]); lin.lora_adapters[key].lora_B.copy_(t[f"B{s}"])
    m = m.to(DEV).eval()
    pkg.calibrate_model(m, bits, [t["x0"].to(DEV), t["x1"].to(DEV)])
    return m, key


@pytest.mark.parametrize("name", ["mlp_mm4", "mlp_mm8", "mlp_log6"])
def test_spmlp_with_fused_gelu(pkg, name):
    meta, t = load_blk(name)
    m, key = build_mlp(pkg, meta, t)
    qt = meta["qtype"]
    tol = 1e-5
    for lin, tag in ((m.c_fc, "fc"), (m.c_proj, "proj")):          # chained calibration (c_proj sees gelu(c_fc)) as the reference's
        got, ref = lin.quantizers_input[key].scale.cpu(), t[f"{tag}.qx.scale"]
        if qt == "minmax" and tag == "fc":
            assert torch.equal(got, ref)
        elif qt == "minmax" or tag == "fc":                         # downstream of a GEMM (+ erf): close, not bit-equal
            assert torch.allclose(got, ref, rtol=2e-5, atol=1e-7), (name, tag)
        else:   # log range of gelu(c_fc): its lower end is log2 of the smallest |h| above eps, where an absolute error of
                # 1e-8 in h is a relative error of 1e-3 -- ill-conditioned in the reference itself
            assert torch.allclose(got, ref, rtol=1e-3, atol=1e-6), (name, tag)
        with torch.no_grad():                                       # compare like for like downstream
            lin.quantizers_input[key].scale = ref.to(DEV); lin.quantizers_input[key].zero_point = t[f"{tag}.qx.zero_point"].to(DEV)
            lin.quantizers_input[key]._epoch += 1
    x2 = t["x2"].to(DEV)
    with torch.no_grad():
        h = m.c_fc(x2, activation="gelu")
        assert m.c_fc._activation_fused, "the GELU did not go into the contraction's store"
        assert_close_y(h, t["h"], f"{name}.gelu(c_fc)", tol)
        h_unfused = torch.nn.functional.gelu(m.c_fc(x2))
        assert_close_y(h, h_unfused.cpu(), f"{name}.fused vs separate gelu", 1e-6)
        y = m(x2)
    # c_proj quantizes h: an element of h within rounding distance of a level boundary may land one level away from the
    # reference's, which moves that output ROW by one quantisation step.  Exact or explained, no allowance: (i) the product's own h
    # through the oracle's c_proj meets the bound on every row; (ii) every row of the end-to-end comparison that is outside the
    # bound holds an element whose oracle level differs between the reference's h and the product's h.
    from oracle import ref_cpu as O
    fc, proj = O.build_calibrated_mlp((t["Wf"], t["bf"], t["Af"], t["Bf"]), (t["Wp"], t["bp"], t["Ap"], t["Bp"]),
                                      [t["x0"], t["x1"]], meta["bits"], qt, True, meta["alpha"], meta["r"])
    proj.qx.scale, proj.qx.zero_point = t["proj.qx.scale"], t["proj.qx.zero_point"]
    assert_close_y(y, proj.forward(h.cpu()), f"{name}.c_proj(h)", tol)
    flipped = (proj.qx.levels(t["h"]) != proj.qx.levels(h.cpu())).any(dim=-1)
    off = rows_outside(y, t["y"], tol)
    assert not bool((off & ~flipped).any()), f"{name}: {int((off & ~flipped).sum())} rows off with no flipped input level of c_proj"
    assert float(flipped.float().mean()) <= 0.05, (name, float(flipped.float().mean()))
    # training mode: separate gelu under autograd, same values
    m.train()
    xg = x2.clone().requires_grad_(True)
    yg = m(xg)
    assert yg.grad_fn is not None and not m.c_fc._activation_fused
    assert_close_y(yg, y.cpu(), f"{name}.train vs eval", 1e-5)


@pytest.mark.parametrize("name", ["block_mm4", "block_mm8"])
def test_spblock_against_reference_fixture(pkg, name):
    """A whole pre-LN block (models_sp.py:130-171) on the drop-ins -- LayerNorm kernel, four fused linears, GELU in c_fc's store,
    stock-torch attention -- against the reference block's output after the same calibration protocol."""
    meta, t = load_blk(name)
    bits, r = meta["bits"], meta["r"]
    cfg = types.SimpleNamespace(n_embd=meta["E"], n_head=meta["H"], n_positions=meta["n_positions"], layer_norm_epsilon=1e-5,
                                bit_widths=[bits, 32], lora_rank_per_bit={bits: r, 32: 0}, lora_alpha_per_bit={bits: meta["alpha"], 32: 0},
                                quantizer_per_bit={bits: meta["qtype"], 32: None}, per_channel_quantization=True)
    blk = pkg.SPBlock(cfg, bit_widths=[bits, 32])
    names = [n for n, _ in blk.named_parameters()]
    assert names == [k[len("param."):] for k in t if k.startswith("param.")], "parameter names / order differ from the reference block"
    with torch.no_grad():
        for n, p_ in blk.named_parameters():
            p_.copy_(t[f"param.{n}"])
    blk = blk.to(DEV).eval()
    x0, x1, x2 = (t[k].to(DEV) for k in ("x0", "x1", "x2"))
    blk.set_precision(32)
    with torch.no_grad():
        assert_close_y(blk(x2), t["y32"], f"{name}.y32", 1e-5)
    pkg.calibrate_model(blk, bits, [x0, x1])
    key = f"{bits}bit"
    lins = [blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc, blk.mlp.c_proj]
    for i, lin in enumerate(lins):
        q = lin.quantizers_input[key]
        assert torch.allclose(q.scale.cpu(), t[f"qx{i}.scale"], rtol=3e-5, atol=1e-7), (name, i)
        with torch.no_grad():
            q.scale = t[f"qx{i}.scale"].to(DEV); q.zero_point = t[f"qx{i}.zero_point"].to(DEV); q._epoch += 1
    with torch.no_grad():
        y = blk(x2)
    assert blk.mlp.c_fc._activation_fused
    # Exact or explained (helpers.cosim_block): every stage of the block meets its bound on EVERY row when the oracle is fed the
    # same input, the real forward is bit-identical to that staged run, and each row of the end-to-end comparison that is outside
    # the bound holds a flipped input level upstream (no statistical allowance).
    for sdpa in (True, False):                      # torch's fused attention kernel, then the reference's explicit formula
        blk.attn.use_sdpa = sdpa
        y_staged, y_oracle, affected, nflip = cosim_block(blk, bits, x2)
        assert_close_y(y_oracle, t["y"], f"{name}: the oracle's block against the reference's", 1e-5)
        off = rows_outside(y_staged, t["y"], 1e-5)
        assert not bool((off.reshape(affected.shape) & ~affected).any()), \
            f"{name} (sdpa={sdpa}): {int((off.reshape(affected.shape) & ~affected).sum())} rows off with no flipped level upstream"
        assert nflip <= 2e-3 * x2.numel() * 8, (name, nflip)        # a handful of ties, not a systematic difference
    blk.attn.use_sdpa = True
    with torch.no_grad():
        assert torch.equal(y, blk(x2))


def _make_block(pkg, name):
    meta, t = load_blk(name)
    bits, r = meta["bits"], meta["r"]
    cfg = types.SimpleNamespace(n_embd=meta["E"], n_head=meta["H"], n_positions=meta["n_positions"], layer_norm_epsilon=1e-5,
                                bit_widths=[bits, 32], lora_rank_per_bit={bits: r, 32: 0}, lora_alpha_per_bit={bits: meta["alpha"], 32: 0},
                                quantizer_per_bit={bits: meta["qtype"], 32: None}, per_channel_quantization=True)
    blk = pkg.SPBlock(cfg, bit_widths=[bits, 32])
    with torch.no_grad():
        for n, p_ in blk.named_parameters():
            p_.copy_(t[f"param.{n}"])
    blk = blk.to(DEV).eval()
    pkg.calibrate_model(blk, bits, [t["x0"].to(DEV), t["x1"].to(DEV)])
    return blk, t, bits


@pytest.mark.parametrize("name", ["block_mm4", "block_mm8"])
def test_layernorm_inside_the_activation_pass_block(pkg, name):
    """ln_1 -> c_attn and ln_2 -> c_fc with the LayerNorm applied inside the consumer's activation pass (the
    normalised tensor is never stored): bit-identical to running the LayerNorm kernel first -- same arithmetic, hence the same
    integer levels -- on the reference-shaped block."""
    blk, t, bits = _make_block(pkg, name)
    lins = [blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc, blk.mlp.c_proj]
    x2 = t["x2"].to(DEV)
    with torch.no_grad():
        for lin in lins:
            lin.fuse_norm = False
        y_sep = blk(x2)
        assert not blk.attn.c_attn._norm_fused and not blk.mlp.c_fc._norm_fused
        for lin in lins:
            lin.fuse_norm = True
        y_fused = blk(x2)
        assert blk.attn.c_attn._norm_fused and blk.mlp.c_fc._norm_fused
        assert torch.equal(y_fused, y_sep)
        # calibration forwards need the normalised tensor itself (statistics): they must not take the prologue
        q = blk.attn.c_attn.quantizers_input[f"{bits}bit"]
        q.start_calibration()
        blk(x2)
        assert not blk.attn.c_attn._norm_fused
        q.finish_calibration()
        # 32-bit teacher path: plain LayerNorm + F.linear
        blk.set_precision(32)
        assert_close_y(blk(x2), t["y32"], f"{name}.y32", 1e-5)


@pytest.mark.parametrize("M,K,N,r,bits,qtype,pc", [
    (8192, 768, 2304, 64, 4, "minmax", True),       # ln_1 -> c_attn of GPT-2-small, 16-row activation kernel
    (32768, 768, 3072, 64, 4, "minmax", True),      # ln_2 -> c_fc at BASELINE config 3's 32 x 1024 tokens: 32-row panel kernel
    (4096, 1024, 4096, 64, 6, "log", True),         # GPT-2-medium dims, log quantizer: two-limb activations
    (4096, 768, 768, 16, 8, "minmax", False),       # per-tensor scale: int8 operand path
    (1000, 256, 512, 0, 4, "minmax", True),         # ragged M, no LoRA
])
def test_layernorm_inside_the_activation_pass_shapes(pkg, M, K, N, r, bits, qtype, pc):
    from llm_qat_on_gpt2_amd import synthetic as S
    W, bias, A, B, x0, x1 = S.make_workload(M, K, N, max(r, 1), seed=K + bits, batch=1)
    ln = pkg.SwitchableLayerNorm(K, precision_levels=[bits, 32], eps=1e-5)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        ln.weights[str(bits)].copy_(torch.randn(K, generator=g) * 0.2 + 1.0); ln.biases[str(bits)].copy_(torch.randn(K, generator=g) * 0.1)
    ln = ln.to(DEV); ln.set_precision(bits)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: max(r, 1), 32: 0}, {bits: qtype, 32: None}, per_channel=pc)
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        if r:
            layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval(); layer.set_precision(bits)
    xin = (x1 * 1.7 + 0.3).to(DEV)
    xin[:, 3] += 25.0                                                   # an outlier channel, as GPT-2 residual streams have
    with torch.no_grad():
        h = ln(xin)
        pkg.calibrate_layer(layer, bits, [h, ln((x0 * 1.7 + 0.3).to(DEV))])
        y_sep = layer(h)
        y_fused = layer(xin, pre_norm=ln)
        assert layer._norm_fused
        assert torch.equal(y_fused, y_sep), f"max abs diff {float((y_fused - y_sep).abs().max()):.3e}"
        y_gelu = layer(xin, activation="gelu", pre_norm=ln)
        assert torch.equal(y_gelu, layer(h, activation="gelu"))
        layer.fuse_norm = False
        assert torch.equal(layer(xin, pre_norm=ln), y_sep) and not layer._norm_fused
