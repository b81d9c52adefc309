"""part2 CPTLinear without a GPU: the oracle against the fixtures the reference produced, and the host
logic of the drop-in classes (state-dict layout, precision switching, error behaviour)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close_y

import llm_qat_on_gpt2_amd as pkg

CPT_CASES = ["log_4_6_8", "minmax_4_8", "log_2_3_5", "log_12_18", "minmax_13_16", "mixed_types", "uncalibrated_6"]


def load_cpt(name, prefix="cpt"):
    z = np.load(os.path.join(GOLDEN, f"{prefix}_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    meta["qpb"] = {int(k): v for k, v in meta["qpb"].items()}
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


@pytest.mark.parametrize("name", CPT_CASES)
def test_cpt_oracle_reproduces_reference_fixtures(name):
    from oracle import ref_cpt as C
    meta, t = load_cpt(name)
    o = C.OracleCPTLayer(t["W"], t["bias"], t["A"], t["B"], meta["widths"], meta["qpb"], rank=meta["r"], alpha=meta["alpha"])
    for b in meta["widths"]:
        if b < 32 and b not in meta["skip_calibration"]:
            o.calibrate(b, [t["x0"], t["x1"]])
    for b in meta["widths"]:
        o.set_precision(b)
        assert_close_y(o.forward(t["x2"]), t[f"y_{b}"], f"{name}.y_{b}")
        if b >= 32:
            continue
        o.calibration_mode = True
        assert_close_y(o.forward(t["x2"]), t[f"base_{b}"], f"{name}.base_{b}")
        o.calibration_mode = False
        if b in meta["skip_calibration"]:
            assert torch.equal(o.q_in(t["x2"]), t["x2"])            # eval-mode pass-through (quantization.py:257-272)
            continue
        for tag, oq, src in (("in", o.q_in, t["x2"]), ("w", o.q_w, t["W"]), ("lora", o.q_lora[b], t["A"])):
            assert torch.equal(oq.scales[b], t[f"{tag}.scale_{b}"]), (name, tag, b)
            assert torch.equal(oq.zero_points[b], t[f"{tag}.zero_point_{b}"]), (name, tag, b)
            assert torch.equal(oq(src), t[f"fq_{tag}_{b}"]), (name, tag, b)
        assert torch.equal(o.q_lora[b](t["B"]), t[f"fq_B_{b}"])


def test_cpt_log_dequantisation_differs_from_part1():
    """The two log quantizers share their levels but not their dequantisation: part1 sends the normalised level through
    * (2^b-1) / (2^b-1) (quantization_methods.py:57,:64), part2 does not (part2 quantization_methods.py:40).  In fp32 that
    round trip is exact at most widths but moves some levels by an ulp at 8, 11, 12 and 18 bits -- hence SPQ_LOG_DIRECT."""
    from oracle import ref_cpt as C, ref_cpu as R
    moved = {}
    for bits in (4, 6, 8, 11, 12, 16, 18):
        n = 2 ** (bits - 1) - 1
        q = torch.arange(-n, n + 1, dtype=torch.float32)
        direct = q / (2 * n) + 0.5
        full = 2 ** bits - 1
        moved[bits] = int((direct != (direct * full) / full).sum())
    assert moved[4] == moved[6] == moved[16] == 0 and moved[8] > 0 and moved[11] > 0 and moved[12] > 0 and moved[18] > 0
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 14, generator=g)
    lo, rng = torch.tensor(-9.0), torch.tensor(11.5)
    for bits in (6, 12):
        assert torch.allclose(R.log_fakequant(x, lo, rng, bits), C.log_fakequant_direct(x, lo, rng, bits), rtol=1e-6)


def test_cpt_state_dict_layout_and_roundtrip():
    cpt = pkg.cpt
    keys = json.load(open(os.path.join(GOLDEN, "cpt_state_dict_keys.json")))["keys"]
    m = cpt.CPTLinear(96, 80, bit_widths=[4, 6, 32], quantizer_per_bit={4: "log", 6: "log", 32: None})
    # the reference's fixture was calibrated at 4 bits only: emulate that state without a GPU
    for q, shape in ((m.quantizer_input, (1, 1, 96)), (m.quantizer_weight, (80, 1)), (m.lora_weight_quantizers["4bit"], (1, 16))):
        q.scales[4] = torch.rand(shape) + 0.5
        q.zero_points[4] = torch.rand(shape)
        q.calibrated_bits.add(4)
    sd = m.state_dict()
    assert sorted(sd.keys()) == keys
    m2 = cpt.CPTLinear(96, 80, bit_widths=[4, 6, 32], quantizer_per_bit={4: "log", 6: "log", 32: None})
    m2.load_state_dict(sd, strict=True)
    assert m2.quantizer_input.calibrated_bits == {4} and torch.equal(m2.quantizer_input.scales[4], m.quantizer_input.scales[4])
    assert torch.equal(m2.lora_weight_quantizers["4bit"].zero_points[4], m.lora_weight_quantizers["4bit"].zero_points[4])
    assert m2.lora_weight_quantizers["6bit"].calibrated_bits == set()
    # a part1-style checkpoint ('scale' / 'zero_point' buffers) is adopted at the active width (quantization.py:97-107)
    sd1 = {k: v for k, v in sd.items() if not k.startswith("quantizer_weight._")}
    sd1["quantizer_weight.scale"], sd1["quantizer_weight.zero_point"] = torch.ones(80, 1), torch.zeros(80, 1)
    m3 = cpt.CPTLinear(96, 80, bit_widths=[4, 6, 32], quantizer_per_bit={4: "log", 6: "log", 32: None})
    m3.load_state_dict(sd1, strict=True)
    assert m3.quantizer_weight.num_bits in m3.quantizer_weight.calibrated_bits


def test_cpt_host_logic():
    cpt = pkg.cpt
    m = cpt.CPTLinear(32, 16, bit_widths=[4, 6, 8, 32], quantizer_per_bit={4: "minmax", 6: "log", 8: "log", 32: None},
                      shared_lora_rank=4, shared_lora_alpha=8)
    assert m.current_bits == 32 and m.quantizer_weight.num_bits == 8 and m.quantizer_weight.quantizer_type == "log"
    assert m.lora_weight_quantizers["4bit"].quantizer_type == "minmax" and m.shared_lora.scaling == 2.0
    assert tuple(m.shared_lora.lora_B.shape) == (16, 4) and tuple(m.shared_lora.lora_A.shape) == (32, 4)
    with pytest.raises(ValueError):
        m.set_precision(5)
    m.set_precision(6)
    assert m.quantizer_input.num_bits == 6 and m.quantizer_weight.num_bits == 6
    # 32 bits: plain F.linear, also on the CPU (cpt_model.py:92-93)
    m.set_precision(32)
    x = torch.randn(3, 32)
    assert torch.equal(m(x), torch.nn.functional.linear(x, m.linear.weight, m.linear.bias))
    # a quantized width has no CPU path
    m.set_precision(6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(x)
    # uncalibrated quantizer: raises while training with gradients on, passes through otherwise (quantization.py:257-272)
    q = cpt.LearnableFakeQuantize(6, quantizer_type="log")
    q.train()
    with pytest.raises(RuntimeError, match="not calibrated for 6-bit"):
        q(torch.randn(4, requires_grad=True))
    q.eval()
    assert torch.equal(q(x), x)
    with torch.no_grad():
        assert torch.equal(q.train()(x), x)
    # GradientQuantizer: identity in both directions while its quantizer is neither collecting nor calibrated
    gq = cpt.LearnableFakeQuantize(8, quantizer_type="minmax", channel_dim=0)
    w = torch.randn(5, 3, requires_grad=True)
    y = cpt.GradientQuantizer.apply(w, gq)
    y.backward(torch.ones_like(y))
    assert torch.equal(w.grad, torch.ones(5, 3))
