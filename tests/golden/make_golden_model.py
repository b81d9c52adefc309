#!/usr/bin/env python3
"""Golden fixtures for BASELINE config 4's caller (a multi-layer SPModel) and for the checkpoint wire format.

The reference is imported from /root/reference (CPU, fp32): a 3-layer `SPLMHeadModel` (models_sp.py:173-458) with random
weights is calibrated by the reference's own `CalibrationManager` (train_sp.py:32-163) at 4-bit minmax and 6-bit log, run on
seeded token ids, saved by the reference's own `save_sp_checkpoints` (deploy.py:125-183), re-loaded the way the evaluation
loader does (main_sp_eval.py:22-78: `per_channel_quantization=False`, `strict=True`) and run again; `convert_to_int8`
(deploy.py:5-62) gives the INT8 export.  Layer 0 keeps the reference's zero-initialised `lora_B` at 6-bit, so its log
quantizers are in the default-fill state (`[r,1]` statistics, quantization.py:164-172,194-197).

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_model.py
Outputs (data only): tests/golden/model_sp3.npz, ckpt_sp3_{4,6}bit.pth, ckpt_sp3_expect.npz"""
import glob
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from transformers import GPT2Config  # noqa: E402
from part1_switchable_precision.models_sp import SPLMHeadModel  # noqa: E402  (the reference)
from part1_switchable_precision.train_sp import CalibrationManager  # noqa: E402
from part1_switchable_precision.deploy import save_sp_checkpoints, convert_to_int8  # noqa: E402

torch.set_grad_enabled(False)

CFG = dict(vocab_size=97, n_positions=32, n_embd=64, n_layer=3, n_head=4, layer_norm_epsilon=1e-5, embd_pdrop=0.0)
BITS = [4, 6, 32]
SP = dict(bit_widths=BITS, lora_rank_per_bit={4: 8, 6: 8, 32: 0}, lora_alpha_per_bit={4: 16, 6: 16, 32: 0},
          quantizer_per_bit={4: "minmax", 6: "log", 32: None}, activation_bits_per_bit={4: 4, 6: 6, 32: 32})


def make_config(per_channel):
    cfg = GPT2Config(**CFG)
    for k, v in SP.items():
        setattr(cfg, k, v)
    cfg.per_channel_quantization = per_channel
    return cfg


def init_weights(model, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if n.endswith("lora_A"):
            continue                                         # keeps the reference's kaiming init (seeded below)
        if n.endswith("lora_B"):
            zero_init = ".h.0." in n and ".6bit." in n       # the reference's init state for layer 0 at 6-bit (lora.py:38)
            p.data = torch.zeros_like(p) if zero_init else torch.randn(p.shape, generator=g) * 0.01
        elif "wte" in n:
            p.data = torch.randn(p.shape, generator=g) * 0.3
        elif "wpe" in n:
            p.data = torch.randn(p.shape, generator=g) * 0.1
        elif ".weights." in n:
            p.data = torch.randn(p.shape, generator=g) * 0.1 + 1.0
        elif p.dim() > 1:
            p.data = torch.randn(p.shape, generator=g) * 0.08
        else:
            p.data = torch.randn(p.shape, generator=g) * 0.1


def main():
    torch.manual_seed(0)
    model = SPLMHeadModel(make_config(True)).eval()
    init_weights(model, 1)
    g = torch.Generator().manual_seed(7)
    calib = [torch.randint(0, CFG["vocab_size"], (3, 24), generator=g) for _ in range(4)]
    ids = torch.randint(0, CFG["vocab_size"], (3, 24), generator=g)
    loader = [{"input_ids": c} for c in calib]
    mgr = CalibrationManager(model, loader, torch.device("cpu"))
    mgr.calibrate_all_precisions(BITS, num_batches=len(calib))
    for b in (4, 6):
        mgr.calibrate_lora_only(b)

    out = {f"param.{n}": p.data.clone() for n, p in model.transformer.named_parameters()}
    out["ids"] = ids
    for i, c in enumerate(calib):
        out[f"calib{i}"] = c
    lin_names = [n for n, m in model.transformer.named_modules() if m.__class__.__name__ == "SPLinearWithLoRA"]
    for b in (4, 6, 32):
        model.set_precision(b)
        y, hs = model.transformer(ids, output_hidden_states=True)
        out[f"y_{b}"] = y
        for i, h in enumerate(hs):
            out[f"h_{b}_{i}"] = h
        if b < 32:
            for n in lin_names:
                q = model.transformer.get_submodule(n).quantizers_input[f"{b}bit"]
                out[f"qx_{b}.{n}.scale"], out[f"qx_{b}.{n}.zero_point"] = q.scale.clone(), q.zero_point.clone()
    zb = model.transformer.h[0].mlp.c_fc.lora_adapters["6bit"].quantize_B
    assert tuple(zb.scale.shape) == (8, 1), zb.scale.shape          # the default-fill shape this fixture is here to carry
    meta = {"cfg": CFG, "sp": {k: ({str(a): c for a, c in v.items()} if isinstance(v, dict) else v) for k, v in SP.items()},
            "linears": lin_names, "n_calib": len(calib)}
    np.savez_compressed(os.path.join(HERE, "model_sp3.npz"), meta=json.dumps(meta), **{k: v.numpy() for k, v in out.items()})
    print("model_sp3: ok; y_4 rms %.3f  y_6 rms %.3f" % (float(out["y_4"].pow(2).mean().sqrt()), float(out["y_6"].pow(2).mean().sqrt())))

    # ---- checkpoint wire format: written by the reference's own save_sp_checkpoints -------------------------------------------
    mc = types.SimpleNamespace(**CFG, **SP, per_channel_quantization=True)
    expect = {"ids": ids}
    with tempfile.TemporaryDirectory() as tmp:
        for m_ in model.modules():                              # never-written buffers (lora.py:42-43,100): make the bytes stable
            for bn in ("weight_quantized", "lora_A_quantized", "lora_B_quantized"):
                if hasattr(m_, bn) and getattr(m_, bn) is not None:
                    getattr(m_, bn).zero_()
        saved = save_sp_checkpoints(model, os.path.join(tmp, "sp_gpt2"), mc)
        assert sorted(saved) == [4, 6]
        for b, path in saved.items():
            # deploy.py:152 writes pickle protocol 4, which torch's weights_only unpickler rejects (opcode FRAME).  The fixture is
            # the file BYTE FOR BYTE as the reference wrote it; the product reads it with its data-only reader
            # (deploy.load_checkpoint_data_only), and so does this script -- no code from the file runs anywhere.
            import shutil
            from llm_qat_on_gpt2_amd.deploy import load_checkpoint_data_only
            shutil.copyfile(path, os.path.join(HERE, f"ckpt_sp3_{b}bit.pth"))
            try:
                torch.load(os.path.join(HERE, f"ckpt_sp3_{b}bit.pth"), map_location="cpu", weights_only=True)
                raise AssertionError("torch's weights_only unpickler read the protocol-4 file: the fixture no longer tests the fallback")
            except Exception as e:  # noqa: BLE001
                assert "Weights only load failed" in str(e) or "Unsupported" in str(e), e
            ck = load_checkpoint_data_only(os.path.join(HERE, f"ckpt_sp3_{b}bit.pth"))
            # the evaluation loader's construction (main_sp_eval.py:22-78, deploy.py:185-253): per-tensor model, strict load
            ev = SPLMHeadModel(make_config(False)).eval()
            ev.set_precision(ck["bit_width"])
            ev.load_state_dict(ck["model_state_dict"], strict=True)
            logits = ev(ids)
            expect[f"logits_{b}"] = logits
            expect[f"hidden_{b}"] = ev.transformer(ids)
            i8 = convert_to_int8(ev)
            for k, v in i8.items():
                expect[f"int8_{b}.{k}"] = v if torch.is_tensor(v) else torch.tensor(v)
            print(f"ckpt {b}-bit: {os.path.getsize(path) / 1e6:.2f} MB, {len(ck['model_state_dict'])} keys, "
                  f"{len(i8)} int8 entries, logits rms {float(logits.pow(2).mean().sqrt()):.3f}")
    np.savez_compressed(os.path.join(HERE, "ckpt_sp3_expect.npz"), meta=json.dumps({"bits": [4, 6]}),
                        **{k: v.numpy() for k, v in expect.items()})


if __name__ == "__main__":
    main()
