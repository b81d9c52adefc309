"""SURVEY.md §8 f1 without a GPU: the oracle's SwitchableLayerNorm / SPMLP against the fixtures the reference produced, and
the host logic of the drop-in classes."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close_y

import llm_qat_on_gpt2_amd as pkg


def load_blk(name):
    z = np.load(os.path.join(GOLDEN, f"blk_{name}.npz"), allow_pickle=False)
    return json.loads(str(z["meta"])), {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


@pytest.mark.parametrize("name", ["ln_768", "ln_1024", "ln_64"])
def test_layernorm_oracle_matches_reference_fixture(name):
    from oracle import ref_cpu as O
    meta, t = load_blk(name)
    for p in (4, 8, 32):
        assert torch.equal(O.switchable_layernorm(t["x"], t[f"w_{p}"], t[f"b_{p}"], 1e-5), t[f"y_{p}"])


@pytest.mark.parametrize("name", ["mlp_mm4", "mlp_mm8", "mlp_log6"])
def test_mlp_oracle_matches_reference_fixture(name):
    from oracle import ref_cpu as O
    meta, t = load_blk(name)
    fc, proj = O.build_calibrated_mlp((t["Wf"], t["bf"], t["Af"], t["Bf"]), (t["Wp"], t["bp"], t["Ap"], t["Bp"]),
                                      [t["x0"], t["x1"]], meta["bits"], meta["qtype"], True, meta["alpha"], meta["r"])
    assert torch.equal(fc.qx.scale, t["fc.qx.scale"]) and torch.equal(proj.qx.scale, t["proj.qx.scale"])
    y, h = O.sp_mlp_forward(t["x2"], fc, proj)
    assert_close_y(h, t["h"], f"{name}.h")
    assert_close_y(y, t["y"], f"{name}.y", 2e-5)


def test_switchable_layernorm_host_logic():
    ln = pkg.SwitchableLayerNorm(32, precision_levels=[8, 4, 32], eps=1e-5)
    assert ln.precision_levels == [4, 8, 32] and ln.current_precision == 32
    assert sorted(ln.state_dict().keys()) == sorted([f"weights.{p}" for p in (4, 8, 32)] + [f"biases.{p}" for p in (4, 8, 32)])
    with pytest.raises(ValueError):
        ln.set_precision(6)
    assert ln.set_precision(8) == 8
    # the ln_layers compatibility view (switchable_batchnorm.py:34-93): .data and .requires_grad reach the parameter
    ln.ln_layers["8"].weight.data = torch.full((32,), 2.0)
    ln.ln_layers["8"].bias.requires_grad = False
    assert torch.equal(ln.weights["8"].data, torch.full((32,), 2.0)) and not ln.biases["8"].requires_grad
    # CPU tensors and autograd take the composed formula (stock torch ops), identical to the reference's
    from oracle import ref_cpu as O
    x = torch.randn(3, 5, 32)
    with torch.no_grad():
        assert torch.equal(ln(x), O.switchable_layernorm(x, ln.weights["8"], ln.biases["8"], 1e-5))
    xg = x.clone().requires_grad_(True)
    ln(xg).sum().backward()
    assert xg.grad is not None and ln.weights["8"].grad is not None


def test_spmlp_host_logic():
    import types
    cfg = types.SimpleNamespace(n_embd=16, bit_widths=[4, 32], lora_rank_per_bit={4: 4, 32: 0}, lora_alpha_per_bit={4: 8, 32: 0},
                                quantizer_per_bit={4: "minmax", 32: None}, per_channel_quantization=True)
    m = pkg.SPMLP(cfg)
    assert isinstance(m.c_fc, pkg.SPLinearWithLoRA) and m.c_fc.out_features == 64 and m.c_proj.in_features == 64
    with pytest.raises(ValueError):
        m.set_precision(8)
    assert m.set_precision(32) == 32
    x = torch.randn(2, 3, 16)
    ref = m.c_proj(torch.nn.functional.gelu(m.c_fc(x)))             # 32-bit path is plain torch and runs anywhere
    assert torch.equal(m(x), ref)
    with pytest.raises(AttributeError):
        pkg.SPMLP(types.SimpleNamespace(n_embd=16))
    with pytest.raises(ValueError):
        m.c_fc(x, activation="relu")
