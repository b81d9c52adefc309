#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (imported from
/root/reference, CPU, fp32) on seeded synthetic inputs, and cross-check the oracle against it bit for bit.

Run in the build container only (the reference does not travel):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Outputs: tests/golden/case_*.npz, tests/golden/config1_checksums.json, tests/golden/state_dict_keys.json.
Fixtures hold data only (inputs, expected buffers/outputs); no reference source is stored.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from part1_switchable_precision.lora import SPLinearWithLoRA  # noqa: E402  (the reference)
from part1_switchable_precision.quantization import LearnableFakeQuantize  # noqa: E402

from oracle import ref_cpu as O  # noqa: E402

torch.set_grad_enabled(False)


def ref_layer(K, N, bits, qtype, per_channel, r, alpha, W, bias, A, B):
    m = SPLinearWithLoRA(K, N, bit_widths=[bits, 32], lora_rank_per_bit={bits: r, 32: 0},
                         lora_alpha_per_bit={bits: alpha, 32: 0},
                         quantizer_per_bit={bits: qtype, 32: None}, per_channel=per_channel)
    m.linear.weight.data.copy_(W)
    m.linear.bias.data.copy_(bias)
    lo = m.lora_adapters[f"{bits}bit"]
    lo.lora_A.data.copy_(A)
    lo.lora_B.data.copy_(B)
    m.set_precision(bits)
    return m


def ref_calibrate(m, bits, calib):
    """train_sp.py:47-123 + 125-163 on one module."""
    key = f"{bits}bit"
    qw = m.quantizers_weight[key]
    qw.start_calibration(); qw(m.linear.weight.data); qw.finish_calibration()
    lo = m.lora_adapters[key]
    lo.quantize_A.start_calibration(); lo.quantize_A(lo.lora_A); lo.quantize_A.finish_calibration()
    lo.quantize_B.start_calibration(); lo.quantize_B(lo.lora_B); lo.quantize_B.finish_calibration()
    qx = m.quantizers_input[key]
    qx.start_calibration()
    m.calibration_mode = True
    for xb in calib:
        m(xb)
    m.calibration_mode = False
    qx.finish_calibration()


def bufs(q):
    return {k: getattr(q, k).clone() for k in ("scale", "zero_point", "running_min", "running_max")}


def eq(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.equal(a, b), f"oracle != reference (bitwise) at {what}: max|d|={float((a-b).abs().max())}"


def layer_case(name, bits, qtype, per_channel, M=96, K=128, N=192, r=16, alpha=16, seed=0, batch=2,
               zero_B=False, edge=False):
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=seed, batch=batch)
    if zero_B:
        B = torch.zeros_like(B)                      # the reference's own init (lora.py:38)
    if edge:
        W[5, :] = 0.0                                # all-zero output channel -> eps clamp
        W[7, :3] = torch.tensor([3e-6, -9.9e-6, 1.01e-5])
        x0[..., 11] = 0.0; x1[..., 11] = 0.0         # all-zero input channel
        x0[0, 0, :4] = torch.tensor([1e-6, -5e-6, 9.99e-6, 1.0001e-5])
        x0[..., 17] *= 1e-4                          # tiny channel
    m = ref_layer(K, N, bits, qtype, per_channel, r, alpha, W, bias, A, B)
    ref_calibrate(m, bits, [x0, x1])
    key = f"{bits}bit"
    x2 = x0 * 1.3 + 0.05                             # an input partly outside the calibrated range
    lo = m.lora_adapters[key]
    out = {"W": W, "bias": bias, "A": A, "B": B, "x0": x0, "x1": x1, "x2": x2}
    for tag, q in (("qx", m.quantizers_input[key]), ("qw", m.quantizers_weight[key]),
                   ("qA", lo.quantize_A), ("qB", lo.quantize_B)):
        for k, v in bufs(q).items():
            out[f"{tag}.{k}"] = v
    out["fq_x2"] = m.quantizers_input[key](x2)
    out["fq_x0"] = m.quantizers_input[key](x0)
    out["fq_W"] = m.quantizers_weight[key](W)
    out["fq_A"] = lo.quantize_A(A)
    out["fq_B"] = lo.quantize_B(B)
    m.calibration_mode = True
    out["base_x2"] = m(x2)
    m.calibration_mode = False
    out["y_x2"] = m(x2)
    out["y_x0"] = m(x0)
    x2d = x2.reshape(-1, K)[:40]                     # 2-D input through a quantizer calibrated on 3-D
    out["y_2d"] = m(x2d)

    # ---- oracle must reproduce the reference bit for bit (elementwise) / closely (GEMMs)
    ol = O.build_calibrated_layer(W, bias, A, B, [x0, x1], bits, qtype, per_channel, alpha, r)
    for tag, q in (("qx", ol.qx), ("qw", ol.qw), ("qA", ol.qA), ("qB", ol.qB)):
        eq(q.scale, out[f"{tag}.scale"], f"{name}.{tag}.scale")
        eq(q.zero_point, out[f"{tag}.zero_point"], f"{name}.{tag}.zero_point")
        eq(q.running_min, out[f"{tag}.running_min"], f"{name}.{tag}.running_min")
        eq(q.running_max, out[f"{tag}.running_max"], f"{name}.{tag}.running_max")
    eq(ol.qx(x2), out["fq_x2"], f"{name}.fq_x2")
    eq(ol.qx(x0), out["fq_x0"], f"{name}.fq_x0")
    eq(ol.qw(W), out["fq_W"], f"{name}.fq_W")
    eq(ol.qA(A), out["fq_A"], f"{name}.fq_A")
    eq(ol.qB(B), out["fq_B"], f"{name}.fq_B")
    eq(ol.forward(x2, calibration_mode=True), out["base_x2"], f"{name}.base_x2")  # same ATen sgemm
    eq(ol.forward(x2), out["y_x2"], f"{name}.y_x2")
    eq(ol.forward(x2d), out["y_2d"], f"{name}.y_2d")
    # integer levels, from the oracle (now known to reproduce the reference's dequantised values)
    out["lv_x2"] = ol.qx.levels(x2).to(torch.int32)
    out["lv_W"] = ol.qw.levels(W).to(torch.int32)
    out["lv_A"] = ol.qA.levels(A).to(torch.int32)
    out["lv_B"] = ol.qB.levels(B).to(torch.int32)
    meta = dict(bits=bits, qtype=qtype, per_channel=per_channel, r=r, alpha=alpha, M=M, K=K, N=N,
                batch=batch, current_bits_default=int(SPLinearWithLoRA(
                    8, 8, [bits, 32], {bits: 2, 32: 0}, {bits: 2, 32: 0},
                    {bits: qtype, 32: None}).current_bits))
    np.savez_compressed(os.path.join(HERE, f"case_{name}.npz"), meta=json.dumps(meta),
                        **{k: v.numpy() for k, v in out.items()})
    print(f"  {name}: ok  y rms={float(out['y_x2'].pow(2).mean().sqrt()):.4f}")


def grad_case(name, bits, qtype, per_channel, M=64, K=96, N=128, r=16, alpha=32, seed=0):
    """Backward of the reference module (base weight frozen as in main_sp.py:83; alpha/rank = 2): grads w.r.t. the input
    and the LoRA factors for a seeded upstream gradient."""
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=seed, batch=2)
    m = ref_layer(K, N, bits, qtype, per_channel, r, alpha, W, bias, A, B)
    ref_calibrate(m, bits, [x0, x1])
    m.linear.weight.requires_grad_(False); m.linear.bias.requires_grad_(False)
    lo = m.lora_adapters[f"{bits}bit"]
    g = torch.randn(2, M // 2, N, generator=torch.Generator().manual_seed(seed + 77)) * (3.0 if qtype == "log" else 1.0)
    xg = (x0 * 1.1).clone().requires_grad_(True)
    with torch.enable_grad():
        y = m(xg)
        y.backward(g)
    out = {"W": W, "bias": bias, "A": A, "B": B, "x0": x0, "x1": x1, "xg": xg.detach(), "g": g, "y": y.detach(),
           "grad_x": xg.grad.clone(), "grad_A": lo.lora_A.grad.clone(), "grad_B": lo.lora_B.grad.clone()}
    ol = O.build_calibrated_layer(W, bias, A, B, [x0, x1], bits, qtype, per_channel, alpha, r)
    gx, gA, gB = O.sp_linear_backward(ol, xg.detach(), g)
    for got, want, what in ((gx, out["grad_x"], "grad_x"), (gA, out["grad_A"], "grad_A"), (gB, out["grad_B"], "grad_B")):
        err = float((got - want).abs().max() / want.abs().max().clamp(min=1e-30))
        assert err < 2e-6, f"oracle backward != reference autograd at {name}.{what}: {err:.2e}"
    meta = dict(bits=bits, qtype=qtype, per_channel=per_channel, r=r, alpha=alpha, M=M, K=K, N=N)
    np.savez_compressed(os.path.join(HERE, f"grad_{name}.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    print(f"  grad {name}: ok")


def quantizer_case(name, bits, qtype, symmetric, per_channel, channel_dim, shape, seed, batches=2,
                   all_tiny_first=False):
    """Standalone LearnableFakeQuantize (covers the asymmetric branches no caller uses, multi-batch
    running stats, and the log 'no element above eps' default branch quantization.py:194-197)."""
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(*shape, generator=g) * (0.5 + i) + 0.1 * i for i in range(batches)]
    if all_tiny_first:
        xs[0] = torch.full(shape, 3e-6)
    q = LearnableFakeQuantize(bits, channel_dim=channel_dim, quantizer_type=qtype, symmetric=symmetric,
                              per_channel=per_channel)
    q.start_calibration()
    for x in xs:
        q(x)
    q.finish_calibration()
    xt = torch.randn(*shape, generator=g) * 1.2
    out = {f"x{i}": x for i, x in enumerate(xs)}
    out["xt"] = xt
    out.update(bufs(q))
    out["fq_xt"] = q(xt)
    oq = O.QuantState(bits, qtype, channel_dim, per_channel, symmetric)
    oq.start()
    for x in xs:
        oq(x)
    oq.finish()
    for k in ("scale", "zero_point", "running_min", "running_max"):
        eq(getattr(oq, k), out[k], f"{name}.{k}")
    eq(oq(xt), out["fq_xt"], f"{name}.fq_xt")
    out["lv_xt"] = oq.levels(xt).to(torch.int32)
    meta = dict(bits=bits, qtype=qtype, symmetric=symmetric, per_channel=per_channel,
                channel_dim=channel_dim, shape=list(shape), batches=batches)
    np.savez_compressed(os.path.join(HERE, f"case_{name}.npz"), meta=json.dumps(meta),
                        **{k: v.numpy() for k, v in out.items()})
    print(f"  {name}: ok")


def config1():
    """BASELINE config 1: 8-bit per-tensor minmax on the full 768x3072 c_fc weight (CPU)."""
    W = O.make_workload(8, 768, 3072, 64, seed=0)[0]
    q = LearnableFakeQuantize(8, channel_dim=0, quantizer_type="minmax", per_channel=False)
    q.start_calibration(); q(W); q.finish_calibration()
    fq = q(W)
    lv = torch.round(fq / q.scale).to(torch.int64)
    oq = O.QuantState(8, "minmax", 0, False).calibrate_on(W)
    eq(oq(W), fq, "config1.fq"); eq(oq.levels(W).to(torch.int64), lv, "config1.levels")
    hist = torch.bincount((lv + 127).flatten(), minlength=255)
    js = {"W_sha256": hashlib.sha256(W.numpy().tobytes()).hexdigest(),
          "scale_shape": list(q.scale.shape), "scale_hex": q.scale.flatten()[0].item().hex(),
          "zero_point": q.zero_point.flatten().tolist(), "level_hist_from_-127": hist.tolist(),
          "distinct_levels": int((hist > 0).sum()), "level_min": int(lv.min()), "level_max": int(lv.max()),
          "level_sum": int(lv.sum()), "level_abs_sum": int(lv.abs().sum()),
          "fq_sum_f64_hex": float(fq.double().sum()).hex(),
          "fq_sha256": hashlib.sha256(fq.numpy().tobytes()).hexdigest(),
          "fq_first64_hex": [v.hex() for v in fq.flatten()[:64].tolist()],
          "fq_last64_hex": [v.hex() for v in fq.flatten()[-64:].tolist()]}
    json.dump(js, open(os.path.join(HERE, "config1_checksums.json"), "w"), indent=1)
    print("  config1: ok distinct levels", js["distinct_levels"], js["level_min"], js["level_max"])


def state_dict_keys():
    m = SPLinearWithLoRA(16, 24, bit_widths=[4, 6, 8, 32], lora_rank_per_bit={4: 4, 6: 4, 8: 4, 32: 0},
                         lora_alpha_per_bit={4: 4, 6: 4, 8: 4, 32: 0},
                         quantizer_per_bit={4: "minmax", 6: "log", 8: "log", 32: None})
    sd = m.state_dict()
    js = {"keys": list(sd.keys()), "shapes": {k: list(v.shape) for k, v in sd.items()},
          "current_bits_default": m.current_bits}
    json.dump(js, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=1)
    print("  state_dict keys:", len(js["keys"]))


if __name__ == "__main__":
    torch.manual_seed(0)
    layer_case("mm4_pc", 4, "minmax", True)
    layer_case("mm8_pt", 8, "minmax", False)
    layer_case("mm8_pc", 8, "minmax", True, seed=3)
    layer_case("mm3_pc", 3, "minmax", True, seed=4)
    layer_case("mm16_pc", 16, "minmax", True, seed=5)
    layer_case("mm4_pc_edge", 4, "minmax", True, seed=6, edge=True)
    layer_case("mm4_pt_zeroB", 4, "minmax", False, seed=7, zero_B=True)
    layer_case("mm4_pc_ragged", 4, "minmax", True, M=70, K=72, N=100, r=8, alpha=16, seed=8, batch=2)
    layer_case("log6_pc", 6, "log", True, seed=9)
    layer_case("log7_pt", 7, "log", False, seed=10)
    layer_case("log8_pc", 8, "log", True, seed=11)
    layer_case("log6_pc_edge", 6, "log", True, seed=12, edge=True)
    layer_case("log6_pc_zeroB", 6, "log", True, seed=13, zero_B=True)
    layer_case("log5_pc_ragged", 5, "log", True, M=70, K=72, N=100, r=8, alpha=4, seed=14, batch=2)
    quantizer_case("q_mm8_asym_pc", 8, "minmax", False, True, 0, (48, 40), 20)
    quantizer_case("q_mm4_asym_pt", 4, "minmax", False, False, 0, (48, 40), 21)
    quantizer_case("q_log6_asym_pc", 6, "log", False, True, 1, (48, 40), 22)
    quantizer_case("q_mm8_sym_mid", 8, "minmax", True, True, 1, (6, 10, 12), 23, batches=3)
    quantizer_case("q_log6_tinyfirst", 6, "log", True, True, -1, (3, 8, 16), 24, all_tiny_first=True)
    quantizer_case("q_log4_pt", 4, "log", True, False, 0, (32, 24), 25)
    with torch.enable_grad():
        torch.set_grad_enabled(True)
        grad_case("mm4_pc", 4, "minmax", True, seed=30)
        grad_case("log6_pc", 6, "log", True, seed=31)
        grad_case("mm8_pt", 8, "minmax", False, seed=32)
        torch.set_grad_enabled(False)
    config1()
    state_dict_keys()
    print("golden fixtures written; oracle == reference bitwise on every elementwise quantity")
