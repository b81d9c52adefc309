#!/usr/bin/env python3
"""Golden fixtures for the block-level pieces of SURVEY.md §8 f1: the reference's SwitchableLayerNorm and SPMLP (imported
from /root/reference, CPU, fp32) on seeded inputs, cross-checked bit for bit (elementwise) against oracle/ref_cpu.py.

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_blocks.py
Outputs: tests/golden/blk_*.npz (data only)."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from part1_switchable_precision.models_sp import SPMLP, SPBlock  # noqa: E402  (the reference)
from part1_switchable_precision.switchable_batchnorm import SwitchableLayerNorm  # noqa: E402

from oracle import ref_cpu as O  # noqa: E402

torch.set_grad_enabled(False)


def ln_case(name, rows_shape, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*rows_shape, C, generator=g) * 1.7 + 0.3
    x[..., 3] += 25.0                                             # an outlier channel, as GPT-2 residual streams have
    ln = SwitchableLayerNorm(C, precision_levels=[4, 8, 32], eps=1e-5)
    out = {"x": x}
    for p in (4, 8, 32):
        ln.weights[str(p)].data = torch.randn(C, generator=g) * 0.2 + 1.0
        ln.biases[str(p)].data = torch.randn(C, generator=g) * 0.1
        ln.set_precision(p)
        y = ln(x)
        yo = O.switchable_layernorm(x, ln.weights[str(p)].data, ln.biases[str(p)].data, 1e-5)
        assert torch.equal(y, yo), f"{name}: oracle != reference at precision {p}"
        out[f"w_{p}"], out[f"b_{p}"], out[f"y_{p}"] = ln.weights[str(p)].data.clone(), ln.biases[str(p)].data.clone(), y
    np.savez_compressed(os.path.join(HERE, f"blk_ln_{name}.npz"), meta=json.dumps({"name": name, "C": C}),
                        **{k: v.numpy() for k, v in out.items()})
    print(f"  ln {name}: ok")


def mlp_case(name, bits, qtype, E=64, r=8, alpha=16, M=96, seed=0):
    cfg = types.SimpleNamespace(n_embd=E, bit_widths=[bits, 32], lora_rank_per_bit={bits: r, 32: 0},
                                lora_alpha_per_bit={bits: alpha, 32: 0}, quantizer_per_bit={bits: qtype, 32: None},
                                per_channel_quantization=True)
    m = SPMLP(cfg, bit_widths=[bits, 32]).eval()
    Wf, bf, Af, Bf, x0, x1 = O.make_workload(M, E, 4 * E, r, seed=seed, batch=2)
    Wp, bp, Ap, Bp, _, _ = O.make_workload(M, 4 * E, E, r, seed=seed + 50, batch=2)
    g = torch.Generator().manual_seed(seed + 9)
    x2 = torch.randn(2, M // 2, E, generator=g) * 1.2
    key = f"{bits}bit"
    for lin, (W, b, A, B) in ((m.c_fc, (Wf, bf, Af, Bf)), (m.c_proj, (Wp, bp, Ap, Bp))):
        lin.linear.weight.data.copy_(W); lin.linear.bias.data.copy_(b)
        lin.lora_adapters[key].lora_A.data.copy_(A); lin.lora_adapters[key].lora_B.data.copy_(B)
    m.set_precision(bits)
    for lin in (m.c_fc, m.c_proj):                      # train_sp.py:58-83, 125-163
        qw = lin.quantizers_weight[key]
        qw.start_calibration(); qw(lin.linear.weight.data); qw.finish_calibration()
        lo = lin.lora_adapters[key]
        lo.quantize_A.start_calibration(); lo.quantize_A(lo.lora_A); lo.quantize_A.finish_calibration()
        lo.quantize_B.start_calibration(); lo.quantize_B(lo.lora_B); lo.quantize_B.finish_calibration()
    for lin in (m.c_fc, m.c_proj):                      # train_sp.py:85-118
        lin.quantizers_input[key].start_calibration(); lin.calibration_mode = True
    for xb in (x0, x1):
        m(xb)
    for lin in (m.c_fc, m.c_proj):
        lin.calibration_mode = False; lin.quantizers_input[key].finish_calibration()
    fc, proj = O.build_calibrated_mlp((Wf, bf, Af, Bf), (Wp, bp, Ap, Bp), [x0, x1], bits, qtype, True, alpha, r)
    for lin, ol, tag in ((m.c_fc, fc, "fc"), (m.c_proj, proj, "proj")):
        assert torch.equal(lin.quantizers_input[key].scale, ol.qx.scale), f"{name}: {tag} input scale"
        assert torch.equal(lin.quantizers_input[key].zero_point, ol.qx.zero_point)
    h = m.act(m.c_fc(x2))
    y = m(x2)
    yo, ho = O.sp_mlp_forward(x2, fc, proj)
    assert torch.allclose(h, ho, rtol=1e-5, atol=1e-6) and torch.allclose(y, yo, rtol=1e-4, atol=1e-5), name
    out = {"Wf": Wf, "bf": bf, "Af": Af, "Bf": Bf, "Wp": Wp, "bp": bp, "Ap": Ap, "Bp": Bp, "x0": x0, "x1": x1, "x2": x2,
           "h": h, "y": y, "fc.qx.scale": m.c_fc.quantizers_input[key].scale, "fc.qx.zero_point": m.c_fc.quantizers_input[key].zero_point,
           "proj.qx.scale": m.c_proj.quantizers_input[key].scale, "proj.qx.zero_point": m.c_proj.quantizers_input[key].zero_point}
    meta = {"name": name, "bits": bits, "qtype": qtype, "E": E, "r": r, "alpha": alpha, "M": M}
    np.savez_compressed(os.path.join(HERE, f"blk_mlp_{name}.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    print(f"  mlp {name}: ok")


def block_case(name, bits, qtype, E=64, H=4, T=24, Bsz=3, r=8, alpha=16, seed=0):
    """A whole reference SPBlock (models_sp.py:130-171) after the CalibrationManager protocol, random weights."""
    cfg = types.SimpleNamespace(n_embd=E, n_head=H, n_positions=32, layer_norm_epsilon=1e-5, bit_widths=[bits, 32],
                                lora_rank_per_bit={bits: r, 32: 0}, lora_alpha_per_bit={bits: alpha, 32: 0},
                                quantizer_per_bit={bits: qtype, 32: None}, per_channel_quantization=True)
    torch.manual_seed(seed)
    blk = SPBlock(cfg, bit_widths=[bits, 32]).eval()
    key = f"{bits}bit"
    g = torch.Generator().manual_seed(seed + 1)
    for p_ in blk.parameters():
        p_.data = torch.randn(p_.shape, generator=g) * (0.02 if p_.dim() > 1 else 0.1) + (1.0 if ("weights" in "") else 0.0)
    for ln in (blk.ln_1, blk.ln_2):
        for k_ in ln.weights:
            ln.weights[k_].data = torch.randn(E, generator=g) * 0.1 + 1.0
    lins = [blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc, blk.mlp.c_proj]
    for lin in lins:
        lin.lora_adapters[key].lora_A.data = O.kaiming_uniform_a5(lin.in_features, r, g)
        lin.lora_adapters[key].lora_B.data = torch.randn(r, lin.out_features, generator=g) * 0.01
    xs = [torch.randn(Bsz, T, E, generator=g) * 1.5 for _ in range(3)]
    blk.set_precision(bits)
    for lin in lins:
        qw = lin.quantizers_weight[key]
        qw.start_calibration(); qw(lin.linear.weight.data); qw.finish_calibration()
        lo = lin.lora_adapters[key]
        lo.quantize_A.start_calibration(); lo.quantize_A(lo.lora_A); lo.quantize_A.finish_calibration()
        lo.quantize_B.start_calibration(); lo.quantize_B(lo.lora_B); lo.quantize_B.finish_calibration()
    for lin in lins:
        lin.quantizers_input[key].start_calibration(); lin.calibration_mode = True
    for xb in xs[:2]:
        blk(xb)
    for lin in lins:
        lin.calibration_mode = False; lin.quantizers_input[key].finish_calibration()
    out = {f"param.{n}": p_.data.clone() for n, p_ in blk.named_parameters()}
    out.update({"x0": xs[0], "x1": xs[1], "x2": xs[2], "y": blk(xs[2])})
    for i, lin in enumerate(lins):
        out[f"qx{i}.scale"] = lin.quantizers_input[key].scale
        out[f"qx{i}.zero_point"] = lin.quantizers_input[key].zero_point
    blk.set_precision(32)
    out["y32"] = blk(xs[2])
    meta = {"name": name, "bits": bits, "qtype": qtype, "E": E, "H": H, "T": T, "r": r, "alpha": alpha, "n_positions": 32}
    np.savez_compressed(os.path.join(HERE, f"blk_block_{name}.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    print(f"  block {name}: ok  y rms {float(out['y'].pow(2).mean().sqrt()):.3f}")


if __name__ == "__main__":
    block_case("mm4", 4, "minmax")
    block_case("mm8", 8, "minmax", seed=3)
    ln_case("768", (2, 40), 768, 0)
    ln_case("1024", (3, 7), 1024, 1)
    ln_case("64", (5,), 64, 2)
    mlp_case("mm4", 4, "minmax")
    mlp_case("mm8", 8, "minmax", seed=1)
    mlp_case("log6", 6, "log", seed=2)
