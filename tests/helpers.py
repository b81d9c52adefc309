"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

LAYER_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_*.npz"))
                     if not os.path.basename(p).startswith("case_q_"))
QUANT_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_q_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"case_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, t


def assert_close_y(y, y_ref, what="y", rel=1e-5):
    """The tolerance of BASELINE.json north_star as made precise in SURVEY.md §7 hard part 4:
    |d| <= 1e-5*|y_ref| + 1e-5*rms(y_ref)  (a pure elementwise rtol fails on the reference itself)."""
    y = y.detach().double().cpu()
    y_ref = y_ref.detach().double().cpu()
    assert y.shape == y_ref.shape, (what, y.shape, y_ref.shape)
    rms = float(y_ref.pow(2).mean().sqrt())
    err = (y - y_ref).abs()
    bound = rel * y_ref.abs() + rel * rms
    worst = float((err / bound).max())
    assert worst <= 1.0, f"{what}: max err/bound = {worst:.3f} (max abs err {float(err.max()):.3e}, rms {rms:.3e})"
    return worst


# ---------------------------------------------------------------------------------------------------------------------
# Chained parity, exact or explained.
#
# A chain of quantized layers cannot be compared end to end at 1e-5: every linear re-quantizes an upstream result that differs
# from the CPU's in the last ulps, and where that result sits on a rounding tie the level -- and with it the token's whole output
# row -- moves by a quantisation step.  `cosim_block` therefore runs the product block STAGE BY STAGE next to the oracle, the
# oracle being fed the product's own input of every stage, and asserts with no allowance at all:
#   * every stage (LayerNorm, c_attn, attention core, c_proj, residual, LayerNorm, c_fc + GELU, c_proj, residual) meets its bound
#     on every row -- the layer bound 1e-5 |y| + 1e-5 rms(y) for the linears and the attention core, 2e-6 for LayerNorm (its
#     reduction order differs from ATen's), bit equality for the residual adds;
#   * the input levels the product computes for every linear are the oracle's levels of the same input (bit-exact for min-max;
#     log: tie-adjacent only);
#   * the staged product run is bit-identical to the block's real (fused) forward.
# It also returns the census that EXPLAINS the end-to-end deviation from the pure oracle chain: the elements whose oracle level
# differs between the pure chain's input and the product's input of the same linear, as token rows (a flip at c_attn's input
# reaches every later token of its sequence through the attention; elsewhere its own row).  The callers assert that every row
# outside the bound of the end-to-end comparison lies in that set.
# ---------------------------------------------------------------------------------------------------------------------
def oracle_layer_of(module, bits):
    """An oracle layer holding exactly what the product module holds under '{bits}bit' (parameters, scales, zero points)."""
    from oracle import ref_cpu as O
    key = f"{bits}bit"

    def qs(q, cd):
        s = O.QuantState(int(q.num_bits), q.quantizer_type, cd, bool(q.per_channel), bool(q.symmetric), float(q.eps))
        s.scale, s.zero_point = q.scale.detach().cpu().clone(), q.zero_point.detach().cpu().clone()
        s.calibrated = True
        return s
    lora = module.lora_adapters[key]
    on = bool(lora.enabled)
    A = lora.lora_A.detach().cpu() if on else None
    B = lora.lora_B.detach().cpu() if on else None
    return O.OracleLayer(module.linear.weight.detach().cpu(), None if module.linear.bias is None else module.linear.bias.detach().cpu(),
                         A, B, qs(module.quantizers_input[key], -1), qs(module.quantizers_weight[key], 0),
                         qs(lora.quantize_A, 1) if on else None, qs(lora.quantize_B, 1) if on else None,
                         float(lora.scaling) if on else 0.0, bits)


def rows_outside(y, ref, tol=1e-5):
    """boolean [rows]: token rows with an element outside tol * |ref| + tol * rms(ref)"""
    yd, yr = y.detach().cpu().double(), ref.detach().cpu().double()
    rms = float(yr.pow(2).mean().sqrt())
    return ((yd - yr).abs() > tol * yr.abs() + tol * rms).any(dim=-1)


def _levels_agree(layer_mod, olayer, x_gpu, bits, what):
    """the product's input levels of x are the oracle's levels of the same x"""
    from oracle import ref_cpu as O
    key = f"{bits}bit"
    got = layer_mod.quantizers_input[key].quantize_levels(x_gpu.contiguous()).cpu().to(torch.float32).reshape(x_gpu.shape)
    q = olayer.qx
    if q.qtype == "minmax":
        want = q.levels(x_gpu.cpu())
        assert torch.equal(got, want), f"{what}: {int((got != want).sum())} input levels differ from the oracle's levels of the same input"
    else:
        want, pre = O.log_levels(x_gpu.cpu(), q.zero_point, q.scale, q.bits, q.symmetric)
        bad = got != want
        if bool(bad.any()):       # only next to a rounding tie (ATen's log2 is a <= 1-ulp kernel, the device's is correctly rounded)
            assert float((got - want).abs()[bad].max()) == 1.0 and float(((pre - torch.floor(pre)) - 0.5).abs()[bad].max()) < 1e-3, what
            assert float(bad.float().mean()) <= 2e-4, what


def cosim_block(blk, bits, x_gpu, affected_in=None, tol=1e-5, x_ref=None):
    """See the comment above.  ``x_ref``: the pure oracle chain's input of this block (default: the product's).  Returns (y of the
    staged product run, y of the pure oracle chain, affected rows [B, T] bool, number of flipped input levels)."""
    import torch.nn.functional as F
    from oracle import ref_cpu as O
    key = f"{bits}bit"
    B_, T_, _ = x_gpu.shape
    lins = {"attn.c_attn": blk.attn.c_attn, "attn.c_proj": blk.attn.c_proj, "mlp.c_fc": blk.mlp.c_fc, "mlp.c_proj": blk.mlp.c_proj}
    ol = {n: oracle_layer_of(m, bits) for n, m in lins.items()}
    ln = {}
    for n, m in (("ln_1", blk.ln_1), ("ln_2", blk.ln_2)):
        ln[n] = (m.weights[str(bits)].detach().cpu(), m.biases[str(bits)].detach().cpu(), float(m.eps))

    def ln_close(p, o, what):
        d = (p.cpu().double() - o.double()).abs()
        assert bool((d <= 2e-6 * o.double().abs() + 2e-6).all()), f"{what}: LayerNorm off by {float(d.max()):.3e}"

    def stage_close(p, o, what):
        off = rows_outside(p, o, tol)
        assert not bool(off.any()), f"{what}: {int(off.sum())} of {off.numel()} rows outside the bound with the SAME input"

    with torch.no_grad():
        x = x_gpu
        p1 = blk.ln_1(x)
        ln_close(p1, O.switchable_layernorm(x.cpu(), *ln["ln_1"]), "ln_1")
        p2 = blk.attn.c_attn(p1)
        _levels_agree(blk.attn.c_attn, ol["attn.c_attn"], p1, bits, "c_attn")
        stage_close(p2, ol["attn.c_attn"].forward(p1.cpu()), "c_attn")
        p3 = blk.attn.core(p2)
        stage_close(p3, O.attention_core(p2.cpu(), blk.attn.n_head), "attention core")
        p4 = blk.attn.c_proj(p3)
        _levels_agree(blk.attn.c_proj, ol["attn.c_proj"], p3, bits, "attn.c_proj")
        stage_close(p4, ol["attn.c_proj"].forward(p3.cpu()), "attn.c_proj")
        x2 = x + p4
        assert torch.equal(x2.cpu(), x.cpu() + p4.cpu()), "residual add"
        p5 = blk.ln_2(x2)
        ln_close(p5, O.switchable_layernorm(x2.cpu(), *ln["ln_2"]), "ln_2")
        p6 = blk.mlp.c_fc(p5, activation="gelu")
        _levels_agree(blk.mlp.c_fc, ol["mlp.c_fc"], p5, bits, "c_fc")
        stage_close(p6, F.gelu(ol["mlp.c_fc"].forward(p5.cpu())), "gelu(c_fc)")
        p7 = blk.mlp.c_proj(p6)
        _levels_agree(blk.mlp.c_proj, ol["mlp.c_proj"], p6, bits, "mlp.c_proj")
        stage_close(p7, ol["mlp.c_proj"].forward(p6.cpu()), "mlp.c_proj")
        y = x2 + p7
        assert torch.equal(y.cpu(), x2.cpu() + p7.cpu()), "residual add"
        assert torch.equal(blk(x_gpu), y), "the block's real (fused) forward differs from its staged run"

        # ---- the census: where does the PURE oracle chain (fed x itself) take another level than the oracle fed the product's input
        xr = x_gpu.cpu() if x_ref is None else x_ref
        r1 = O.switchable_layernorm(xr, *ln["ln_1"])
        r2 = ol["attn.c_attn"].forward(r1)
        r3 = O.attention_core(r2, blk.attn.n_head)
        r4 = ol["attn.c_proj"].forward(r3)
        xr2 = xr + r4
        r5 = O.switchable_layernorm(xr2, *ln["ln_2"])
        r6 = F.gelu(ol["mlp.c_fc"].forward(r5))
        aff = torch.zeros(B_, T_, dtype=torch.bool) if affected_in is None else affected_in.clone()
        nflip = 0

        def flips(name, x_ref, x_prod):
            f = (ol[name].qx.levels(x_ref) != ol[name].qx.levels(x_prod.cpu())).any(dim=-1)
            return f, int((ol[name].qx.levels(x_ref) != ol[name].qx.levels(x_prod.cpu())).sum())
        f, n = flips("attn.c_attn", r1, p1); nflip += n
        aff = aff | f
        aff = torch.cummax(aff.to(torch.int8), dim=1).values.bool()          # through the causal attention: every later token
        for name, xr_, xp_ in (("attn.c_proj", r3, p3), ("mlp.c_fc", r5, p5), ("mlp.c_proj", r6, p6)):
            f, n = flips(name, xr_, xp_); nflip += n
            aff = aff | f
        y_ref = xr2 + ol["mlp.c_proj"].forward(r6)
    return y, y_ref, aff, nflip
