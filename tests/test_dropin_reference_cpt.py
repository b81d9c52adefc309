"""CPU, build container only (skipped where /root/reference is absent): part2's own model assembly
(part2_cyclic_precision_training/cpt_model.py: CPTModel -> CPTBlock -> CPTSelfAttention / CPTLinear) built on this build's
CPTLinear, the one-name swap of INTEGRATION.md.  No quantized compute (no GPU here)."""
import importlib
import os
import sys
import types

import pytest
import torch

REF = "/root/reference/part2_cyclic_precision_training"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (only the build container has it)")


def _config():
    model = types.SimpleNamespace(vocab_size=97, n_positions=32, n_embd=32, n_layer=2, n_head=4, layer_norm_epsilon=1e-5,
                                  embd_pdrop=0.0, bit_widths=[4, 6, 32], shared_lora_rank=4, shared_lora_alpha=8,
                                  quantizer_per_bit={4: "log", 6: "log", 32: None}, gradient_bits=8)
    return {"model": model, "training": types.SimpleNamespace(target_bits=6)}


def _build(swap):
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)       # the reference imports its siblings by bare name (quantization, quantization_methods)
    for name in ("cpt_model", "quantization", "quantization_methods"):
        sys.modules.pop(name, None)
    cm = importlib.import_module("cpt_model")
    if swap:
        import llm_qat_on_gpt2_amd as pkg
        cm.CPTLinear = pkg.cpt.CPTLinear
    torch.manual_seed(0)
    return cm.CPTModel(_config())


def test_reference_cpt_model_builds_on_the_dropin_layer():
    import llm_qat_on_gpt2_amd as pkg
    try:
        ref, mine = _build(False), _build(True)
    finally:
        sys.path.remove(REF)
        for name in ("cpt_model", "quantization", "quantization_methods"):
            sys.modules.pop(name, None)
    layers = [m for m in mine.modules() if m.__class__.__name__ == "CPTLinear"]
    assert len(layers) == 2 * 4 + 1 and all(isinstance(m, pkg.cpt.CPTLinear) for m in layers)      # 4 per block + lm_head
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    assert list(sd_ref.keys()) == list(sd_mine.keys())
    assert {k: tuple(v.shape) for k, v in sd_ref.items()} == {k: tuple(v.shape) for k, v in sd_mine.items()}
    mine.load_state_dict(sd_ref, strict=True)
    mine.set_precision(4)
    assert all(m.current_bits == 4 and m.quantizer_input.num_bits == 4 for m in layers)
    mine.disable_lora_for_calibration()
    assert all(m.calibration_mode for m in layers)
    mine.enable_lora_after_calibration()
    assert not any(m.calibration_mode for m in layers)
    # the 32-bit path is plain F.linear and runs anywhere: same logits as the reference model with the same weights
    ref.set_precision(32); mine.set_precision(32); ref.eval(); mine.eval()
    ids = torch.randint(0, 97, (2, 16))
    with torch.no_grad():
        a, b = ref(ids), mine(ids)
    assert torch.equal(a.logits, b.logits)
