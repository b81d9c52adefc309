#!/usr/bin/env python3
"""Golden fixtures for part2's CPTLinear: RUN THE REFERENCE (imported from
/root/reference/part2_cyclic_precision_training, CPU, fp32) on seeded synthetic inputs and cross-check
oracle/ref_cpt.py against it bit for bit.

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cpt.py
Outputs: tests/golden/cpt_*.npz, tests/golden/cpt_state_dict_keys.json (data only; no reference source is stored).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/part2_cyclic_precision_training")   # the reference imports its siblings by bare name
sys.dont_write_bytecode = True

from cpt_model import CPTLinear  # noqa: E402  (the reference)

from oracle import ref_cpt as C  # noqa: E402

torch.set_grad_enabled(False)


def ref_calibrate(m, bits, batches):
    """calibration.py:17-88 and :161-203 on one module."""
    m.set_precision(bits)
    qw = m.quantizer_weight
    qw.set_num_bits(bits); qw.start_calibration(); qw(m.linear.weight.data); qw.finish_calibration(debug=False)
    qi = m.quantizer_input
    qi.set_num_bits(bits); qi.start_calibration()
    m.calibration_mode = True
    for xb in batches:
        m(xb)
    m.calibration_mode = False
    qi.finish_calibration(debug=False)
    ql = m.lora_weight_quantizers[f"{bits}bit"]
    ql.set_num_bits(bits); ql.start_calibration(); ql(m.shared_lora.lora_A); ql(m.shared_lora.lora_B)
    ql.finish_calibration(debug=False)


def same(a, b, what):
    assert a.shape == b.shape and torch.equal(a, b), f"oracle != reference at {what}"


def layer_case(name, widths, qpb, M=48, K=96, N=80, r=16, alpha=32, seed=0, skip_calibration=()):
    W, bias, A, B, x0, x1 = C.make_cpt_workload(M, K, N, r, seed=seed, batch=2)
    g = torch.Generator().manual_seed(seed + 77)
    x2 = torch.randn(2, M // 2, K, generator=g) * 1.3
    m = CPTLinear(K, N, bit_widths=widths, quantizer_per_bit=qpb, shared_lora_rank=r, shared_lora_alpha=alpha).eval()
    m.linear.weight.data.copy_(W); m.linear.bias.data.copy_(bias)
    m.shared_lora.lora_A.data.copy_(A); m.shared_lora.lora_B.data.copy_(B)
    o = C.OracleCPTLayer(W, bias, A, B, widths, qpb, rank=r, alpha=alpha)
    out = {"W": W, "bias": bias, "A": A, "B": B, "x0": x0, "x1": x1, "x2": x2}
    student = [b for b in widths if b < 32]
    for b in student:
        if b in skip_calibration:
            continue
        ref_calibrate(m, b, [x0, x1])
        o.calibrate(b, [x0, x1])
    for b in widths:
        m.set_precision(b); o.set_precision(b)
        y, yo = m(x2), o.forward(x2)
        assert torch.allclose(y, yo, rtol=1e-5, atol=1e-6), f"{name}: forward at {b} bits"
        out[f"y_{b}"] = y
        if b >= 32:
            continue
        m.calibration_mode = o.calibration_mode = True
        out[f"base_{b}"] = m(x2)
        assert torch.allclose(out[f"base_{b}"], o.forward(x2), rtol=1e-5, atol=1e-6)
        m.calibration_mode = o.calibration_mode = False
        if b in skip_calibration:
            assert b not in m.quantizer_input.calibrated_bits
            continue
        for tag, q, oq, t in (("in", m.quantizer_input, o.q_in, x2), ("w", m.quantizer_weight, o.q_w, W),
                              ("lora", m.lora_weight_quantizers[f"{b}bit"], o.q_lora[b], A)):
            same(q.scales[b], oq.scales[b], f"{name}.{tag}.scale[{b}]")
            same(q.zero_points[b], oq.zero_points[b], f"{name}.{tag}.zp[{b}]")
            out[f"{tag}.scale_{b}"] = q.scales[b]
            out[f"{tag}.zero_point_{b}"] = q.zero_points[b]
            fq = q(t)
            same(fq, oq(t), f"{name}.{tag}.fq[{b}]")
            out[f"fq_{tag}_{b}"] = fq
        out[f"fq_B_{b}"] = m.lora_weight_quantizers[f"{b}bit"](B)
        same(out[f"fq_B_{b}"], o.q_lora[b](B), f"{name}.fq_B[{b}]")
    meta = {"name": name, "M": M, "K": K, "N": N, "r": r, "alpha": alpha, "widths": widths,
            "qpb": {str(k): v for k, v in qpb.items()}, "skip_calibration": list(skip_calibration)}
    np.savez_compressed(os.path.join(HERE, f"cpt_{name}.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    print(f"  cpt {name}: ok")
    return m


def grad_case(name, widths, qpb, bits, grad_quantizers, M=64, K=96, N=128, r=16, alpha=32, seed=0):
    """Backward of the reference module under autograd (base weight frozen, LoRA factors and the input trainable).
    grad_quantizers: calibrate the adapter's gradient quantizers on one backward first (their statistics pass returns
    the gradient unchanged), so the second backward fake-quantizes d/dA and d/dB (quantization.py:14-26)."""
    W, bias, A, B, x0, x1 = C.make_cpt_workload(M, K, N, r, seed=seed, batch=2)
    m = CPTLinear(K, N, bit_widths=widths, quantizer_per_bit=qpb, shared_lora_rank=r, shared_lora_alpha=alpha).train()
    m.linear.weight.data.copy_(W); m.linear.bias.data.copy_(bias)
    m.shared_lora.lora_A.data.copy_(A); m.shared_lora.lora_B.data.copy_(B)
    ref_calibrate(m, bits, [x0, x1])
    m.set_precision(bits)
    m.linear.weight.requires_grad_(False); m.linear.bias.requires_grad_(False)
    gen = torch.Generator().manual_seed(seed + 5)
    xg = torch.randn(2, M // 2, K, generator=gen)
    g = torch.randn(2, M // 2, N, generator=gen) * 0.05
    lo = m.shared_lora
    out = {"W": W, "bias": bias, "A": A, "B": B, "x0": x0, "x1": x1, "xg": xg, "g": g}
    with torch.enable_grad():
        if grad_quantizers:
            lo.grad_quantizer_A.start_calibration(); lo.grad_quantizer_B.start_calibration()
            x_ = xg.clone().requires_grad_(True)
            m(x_).backward(g)
            lo.grad_quantizer_A.finish_calibration(); lo.grad_quantizer_B.finish_calibration()
            assert 8 in lo.grad_quantizer_A.calibrated_bits and 8 in lo.grad_quantizer_B.calibrated_bits
            out["gqA.scale"], out["gqB.scale"] = lo.grad_quantizer_A.scales[8], lo.grad_quantizer_B.scales[8]
            out["grad_A_unquantized"], out["grad_B_unquantized"] = lo.lora_A.grad.clone(), lo.lora_B.grad.clone()
            lo.lora_A.grad = None; lo.lora_B.grad = None
        x_ = xg.clone().requires_grad_(True)
        y = m(x_)
        y.backward(g)
    out.update({"y": y.detach(), "grad_x": x_.grad, "grad_A": lo.lora_A.grad, "grad_B": lo.lora_B.grad})
    meta = {"name": name, "M": M, "K": K, "N": N, "r": r, "alpha": alpha, "widths": widths, "bits": bits,
            "qpb": {str(k): v for k, v in qpb.items()}, "grad_quantizers": grad_quantizers}
    np.savez_compressed(os.path.join(HERE, f"cptgrad_{name}.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    print(f"  cpt grad {name}: ok")


def main():
    grad_case("log6", [4, 6, 32], {4: "log", 6: "log", 32: None}, 6, False)
    grad_case("minmax8_gq", [4, 8, 32], {4: "minmax", 8: "minmax", 32: None}, 8, True, seed=1)
    grad_case("log4_gq", [4, 6, 32], {4: "log", 6: "log", 32: None}, 4, True, seed=2)
    layer_case("log_4_6_8", [4, 6, 8, 32], {4: "log", 6: "log", 8: "log", 32: None})
    layer_case("minmax_4_8", [4, 8, 32], {4: "minmax", 8: "minmax", 32: None}, seed=1)
    layer_case("log_2_3_5", [2, 3, 5, 32], {2: "log", 3: "log", 5: "log", 32: None}, seed=2, r=8, alpha=16, K=72, N=52)
    layer_case("log_12_18", [12, 18, 32], {12: "log", 18: "log", 32: None}, seed=3)
    layer_case("minmax_13_16", [13, 16, 32], {13: "minmax", 16: "minmax", 32: None}, seed=4)
    # quantizer type of the largest student width rules the weight/input quantizers (cpt_model.py:70-78)
    layer_case("mixed_types", [4, 6, 32], {4: "minmax", 6: "log", 32: None}, seed=5)
    # eval-mode pass-through of an uncalibrated width (quantization.py:257-272)
    m = layer_case("uncalibrated_6", [4, 6, 32], {4: "log", 6: "log", 32: None}, seed=6, skip_calibration=(6,))
    keys = sorted(m.state_dict().keys())
    json.dump({"keys": keys}, open(os.path.join(HERE, "cpt_state_dict_keys.json"), "w"), indent=1)
    print(f"  state_dict: {len(keys)} keys")


if __name__ == "__main__":
    main()
