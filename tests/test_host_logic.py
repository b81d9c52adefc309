"""CPU: host-side logic of the drop-in classes -- constructor surface, state machine, state-dict layout and the
variable-shape checkpoint hook -- without any compute call (no GPU here)."""
import json
import os

import pytest
import torch

from helpers import GOLDEN

import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd.fake_quantize import _chan_view, _param_axis


def make_layer(bit_widths=(4, 6, 8, 32), K=16, N=24, r=4):
    bw = list(bit_widths)
    return pkg.SPLinearWithLoRA(K, N, bit_widths=bw, lora_rank_per_bit={b: (r if b < 32 else 0) for b in bw},
                                lora_alpha_per_bit={b: (r if b < 32 else 0) for b in bw},
                                quantizer_per_bit={4: "minmax", 6: "log", 8: "log", 32: None})


def test_state_dict_layout_matches_reference():
    """Key list, order and shapes of a freshly built module equal the reference's (golden JSON from the reference)."""
    js = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    layer = make_layer()
    sd = layer.state_dict()
    assert list(sd.keys()) == js["keys"] and len(sd) == 63
    assert {k: list(v.shape) for k, v in sd.items()} == js["shapes"]
    assert layer.current_bits == js["current_bits_default"] == 8      # second-largest width (lora.py:71)
    assert "input_quantized" not in sd                                 # a None buffer
    assert layer.__class__.__name__ == "SPLinearWithLoRA"             # matched by name in models_sp.py:239


def test_constructor_surface_and_defaults():
    layer = make_layer()
    assert set(layer.quantizers_weight.keys()) == {"4bit", "6bit", "8bit"} == set(layer.quantizers_input.keys())
    qw, qx = layer.quantizers_weight["6bit"], layer.quantizers_input["6bit"]
    assert (qw.channel_dim, qw.is_input, qw.quantizer_type, qw.num_bits) == (0, False, "log", 6)
    assert (qx.channel_dim, qx.is_input, qx.symmetric, qx.per_channel) == (-1, True, True, True)
    assert (qx.quant_min, qx.quant_max) == (-32, 31)
    lo = layer.lora_adapters["4bit"]
    assert lo.enabled and lo.scaling == 1.0 and lo.quantize_A.channel_dim == 1 and lo.quantize_B.channel_dim == 1
    assert tuple(lo.lora_A.shape) == (16, 4) and tuple(lo.lora_B.shape) == (4, 24) and float(lo.lora_B.detach().abs().sum()) == 0.0
    assert layer.calibration_mode is False and layer.get_active_lora() is layer.lora_adapters["8bit"]
    pt = pkg.LearnableFakeQuantize(8, channel_dim=0, per_channel=False)
    assert pt.channel_dim is None
    off = pkg.LoRALayer(8, 8, rank=0, alpha=0, bits=32, quantizer_type=None)
    assert not off.enabled and off.quantize_A is None and tuple(off.lora_A.shape) == (1, 1)
    assert tuple(off(torch.zeros(2, 3, 8)).shape) == (2, 3, 8)       # disabled adapter: zeros, no kernel


def test_set_precision_is_attribute_flips_only(capsys):
    layer = make_layer()
    for q in (layer.quantizers_weight["4bit"], layer.quantizers_input["4bit"]):
        q.calibrated = True
    assert layer.set_precision(4) == 4 and layer.current_bits == 4
    assert layer.quantizers_weight["4bit"].calibrated                  # same width: calibration kept (quantization.py:82)
    assert layer.set_precision(32) == 32 and layer.current_bits == 32
    with pytest.raises(KeyError):
        layer.set_precision(5)
    q = layer.quantizers_weight["4bit"]
    q.set_num_bits(5)
    assert not q.calibrated and "Reset calibration" in capsys.readouterr().out
    with pytest.raises(IndexError):
        pkg.SPLinearWithLoRA(8, 8, [4], {4: 2}, {4: 2}, {4: "minmax"})


def test_calibration_state_machine_without_compute():
    q = pkg.LearnableFakeQuantize(8)
    assert not q.calibrated and not q.collecting_stats
    q.start_calibration()
    assert q.collecting_stats and q.num_batches_collected == 0 and q.temp_min is None
    q.finish_calibration()                                              # nothing collected: stays uncalibrated
    assert not q.calibrated and not q.collecting_stats
    with pytest.raises(RuntimeError, match="not calibrated"):
        q(torch.zeros(2, 2))
    q.start_calibration()
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # statistics are a HIP kernel; CPU tensors refused
        q(torch.zeros(2, 2))
    assert pkg.LearnableFakeQuantize(32)(torch.ones(2)) is not None     # >= 32 bits: identity, no kernel


def test_load_state_dict_resizes_variable_shape_buffers():
    """quantization.py:40-75: buffers take the incoming shape, `calibrated` flips on, legacy [*,T,*] input stats collapse."""
    layer = make_layer()
    sd = layer.state_dict()
    sd["quantizers_weight.4bit.scale"] = torch.full((24, 1), 0.5)
    sd["quantizers_weight.4bit.zero_point"] = torch.zeros(24, 1)
    sd["quantizers_weight.4bit.running_min"] = torch.full((24, 1), -1.0)
    sd["quantizers_weight.4bit.running_max"] = torch.full((24, 1), 1.0)
    legacy = torch.arange(3 * 16, dtype=torch.float32).reshape(1, 3, 16)
    for k in ("scale", "zero_point", "running_min", "running_max"):
        sd[f"quantizers_input.4bit.{k}"] = legacy.clone()
    layer2 = make_layer()
    layer2.load_state_dict(sd, strict=True)
    qw, qx = layer2.quantizers_weight["4bit"], layer2.quantizers_input["4bit"]
    assert tuple(qw.scale.shape) == (24, 1) and qw.calibrated and float(qw.scale[3, 0]) == 0.5
    assert tuple(qx.scale.shape) == (1, 1, 16) and qx.calibrated
    assert torch.equal(qx.running_min, legacy.min(dim=1, keepdim=True)[0])       # minmax: min for *min*
    assert torch.equal(qx.running_max, legacy.max(dim=1, keepdim=True)[0])
    assert torch.equal(qx.scale, legacy.max(dim=1, keepdim=True)[0])
    e0 = qw._epoch
    layer2.load_state_dict(sd)
    assert qw._epoch > e0                                                         # prepared operands are invalidated


def test_views_and_broadcast_axes():
    assert _chan_view((8, 1024, 768), 2) == (8 * 1024, 768, 1)         # input, channel_dim=-1
    assert _chan_view((3072, 768), 0) == (1, 3072, 768)                # weight, channel_dim=0
    assert _chan_view((768, 64), 1) == (768, 64, 1)                    # LoRA A
    assert _chan_view((5, 7), None) == (1, 1, 35)
    assert _param_axis((4, 9, 768), (1, 1, 768)) == 2
    assert _param_axis((9, 768), (1, 1, 768)) == 1                     # 3-D keep-dim scale against a 2-D input
    assert _param_axis((3072, 768), (3072, 1)) == 0
    assert _param_axis((3072, 768), (1, 1)) is None
    with pytest.raises(RuntimeError):
        _param_axis((4, 9, 700), (1, 1, 768))
    with pytest.raises(ValueError):
        _param_axis((4, 9), (4, 9))


def test_operand_path_choice():
    layer = make_layer()
    qx4, qx6 = layer.quantizers_input["4bit"], layer.quantizers_input["6bit"]
    L = pkg._lib
    assert layer._choose_path(qx4, None, layer.lora_adapters["4bit"], True, 1) == L.PATH_F16X2
    assert layer._choose_path(qx4, None, layer.lora_adapters["4bit"], True, 0) == L.PATH_F32    # calibration: raw x
    assert layer._choose_path(qx6, None, layer.lora_adapters["6bit"], True, 1) == L.PATH_F16X3  # log input quantizer: limbs
    layer.operand_path = L.PATH_F32
    assert layer._choose_path(qx4, None, layer.lora_adapters["4bit"], True, 1) == L.PATH_F32
    layer.operand_path = L.PATH_F16X2
    assert layer._choose_path(qx6, None, layer.lora_adapters["6bit"], True, 1) == L.PATH_F32    # pinned but invalid
    layer.operand_path = L.PATH_F32
    assert layer._choose_path(qx6, None, layer.lora_adapters["6bit"], True, 1) == L.PATH_F32
    # int8 matrix cores: only when the input scale is per tensor and the weights are symmetric minmax <= 8 bit; AUTO takes it
    qw4 = layer.quantizers_weight["4bit"]
    layer.operand_path = L.PATH_I8
    qx4.scale = torch.ones(1, 1, layer.in_features)
    assert layer._choose_path(qx4, qw4, layer.lora_adapters["4bit"], True, 1) == L.PATH_F16X2     # per-channel: fp16 limbs
    assert layer._choose_path(qx6, layer.quantizers_weight["6bit"], layer.lora_adapters["6bit"], True, 1) == L.PATH_F16X3
    qx4.scale = torch.ones(1, 1, 1)
    assert layer._choose_path(qx4, qw4, layer.lora_adapters["4bit"], True, 1) == L.PATH_I8
    assert layer._choose_path(qx4, qw4, layer.lora_adapters["4bit"], True, 0) == L.PATH_F32       # calibration forward: raw x
    layer.operand_path = L.PATH_AUTO
    assert layer._choose_path(qx4, qw4, layer.lora_adapters["4bit"], True, 1) == L.PATH_I8


def test_forward_refuses_cpu_tensors_and_teacher_path_is_plain_linear():
    layer = make_layer()
    x = torch.randn(2, 3, 16)
    layer.set_precision(32)
    assert torch.equal(layer(x), torch.nn.functional.linear(x, layer.linear.weight, layer.linear.bias))
    layer.set_precision(4)
    for q in (layer.quantizers_weight["4bit"], layer.quantizers_input["4bit"]):
        q.calibrated = True
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(x)
