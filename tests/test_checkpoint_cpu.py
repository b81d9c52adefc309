"""CPU: the checkpoint wire format -- the files of tests/golden are BYTE FOR BYTE what the reference's own
``save_sp_checkpoints`` wrote (deploy.py:143-152, pickle protocol 4; tests/golden/make_golden_model.py).  They load with
``strict=True`` into this build's ``SPLMHeadModel`` built the way the evaluation loader builds it (main_sp_eval.py:22-78), through
``deploy.load_checkpoint_data_only`` -- torch's ``weights_only`` unpickler refuses protocol 4 -- which runs nothing from the
file.  No compute."""
import collections
import os
import pickle

import pytest
import torch

from helpers import GOLDEN


@pytest.mark.parametrize("bits", [4, 6])
def test_reference_checkpoint_loads_strict(bits):
    from llm_qat_on_gpt2_amd import deploy
    import llm_qat_on_gpt2_amd as pkg
    model, ck = deploy.load_sp_checkpoint(os.path.join(GOLDEN, f"ckpt_sp3_{bits}bit.pth"), device="cpu")
    assert ck["bit_width"] == bits and model.get_current_precision() == bits
    sd = model.state_dict()
    assert list(sd.keys()) == list(ck["model_state_dict"].keys())
    for k, v in ck["model_state_dict"].items():
        assert sd[k].shape == v.shape, k
        if not k.endswith("_quantized"):
            assert torch.equal(sd[k], v), k
    layers = [m for m in model.modules() if isinstance(m, pkg.SPLinearWithLoRA)]
    assert len(layers) == 12
    for m in layers:
        assert m.current_bits == bits
        for b in (4, 6):
            key = f"{b}bit"
            assert m.quantizers_weight[key].calibrated and m.quantizers_input[key].calibrated
            assert tuple(m.quantizers_weight[key].scale.shape) == (m.out_features, 1)      # per-channel buffers, per-tensor model
            assert tuple(m.quantizers_input[key].scale.shape) == (1, 1, m.in_features)
            assert m.lora_adapters[key].quantize_A.calibrated and m.lora_adapters[key].quantize_B.calibrated
    # layer 0 at 6 bit: the reference's zero-initialised lora_B under a log quantizer -> [r,1] default-fill statistics
    qb = model.transformer.h[0].mlp.c_fc.lora_adapters["6bit"].quantize_B
    assert tuple(qb.scale.shape) == (8, 1) and tuple(qb.running_min.shape) == (8, 1)
    s, z = qb.qparams_for(256)
    assert s.numel() == 1 and z.numel() == 1 and float(s) == 0.0 and abs(float(z) - (-16.609640)) < 1e-5
    qb1 = model.transformer.h[1].mlp.c_fc.lora_adapters["6bit"].quantize_B
    assert tuple(qb1.scale.shape) == (1, 256) and qb1.qparams_for(256)[0] is qb1.scale


@pytest.mark.parametrize("bits", [4, 6])
def test_fixture_is_the_reference_wire_format(bits):
    """the fixture really is a protocol-4 file that torch's safe loader cannot read, and the data-only reader reads it whole"""
    from llm_qat_on_gpt2_amd import deploy
    path = os.path.join(GOLDEN, f"ckpt_sp3_{bits}bit.pth")
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)
    ck = deploy.load_checkpoint_data_only(path)
    assert ck["bit_width"] == bits and isinstance(ck["model_state_dict"], collections.OrderedDict) and len(ck["model_state_dict"]) == 564
    assert all(torch.is_tensor(v) and v.device.type == "cpu" for v in ck["model_state_dict"].values())
    assert ck["model_config"]["quantizer_per_bit"] == {4: "minmax", 6: "log", 32: None}


def test_data_only_reader_round_trip_and_refusals(tmp_path):
    from llm_qat_on_gpt2_amd import deploy
    sd = collections.OrderedDict(a=torch.randn(3, 4), b=torch.arange(5), c=torch.nn.Parameter(torch.randn(2)), e=torch.zeros(0),
                                 v=torch.randn(6)[1:4], h=torch.randn(4).half(), t=torch.tensor([True, False]))
    ck = {"model_state_dict": sd, "model_config": {"x": 1, "l": [1, 2], "d": {4: "minmax", 32: None}}, "bit_width": 4, "training_config": None}
    for proto in (2, 4, 5):
        f = str(tmp_path / f"p{proto}.pth")
        torch.save(ck, f, pickle_protocol=proto)
        r = deploy.load_checkpoint_data_only(f)
        assert list(r["model_state_dict"]) == list(sd) and r["model_config"] == ck["model_config"] and r["training_config"] is None
        for k in sd:
            assert r["model_state_dict"][k].dtype == sd[k].dtype and torch.equal(r["model_state_dict"][k], sd[k].detach()), k
        assert isinstance(r["model_state_dict"]["c"], torch.nn.Parameter)

    class Evil:                                              # anything that is not plain data is refused, nothing runs
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    g = str(tmp_path / "evil.pth")
    torch.save({"model_state_dict": sd, "x": Evil()}, g, pickle_protocol=4)
    with pytest.raises(pickle.UnpicklingError, match="no business"):
        deploy.load_checkpoint_data_only(g)
    with pytest.raises(pickle.UnpicklingError):
        deploy.load_sp_checkpoint(g, device="cpu")
    assert not (tmp_path / "pwned").exists()
    with open(tmp_path / "junk.pth", "wb") as fh:
        fh.write(b"not a zip")
    with pytest.raises(Exception):
        deploy.load_checkpoint_data_only(str(tmp_path / "junk.pth"))


def test_qparams_for_refuses_non_uniform_odd_shapes():
    import llm_qat_on_gpt2_amd as pkg
    q = pkg.LearnableFakeQuantize(6, channel_dim=1, quantizer_type="log")
    q.scale = torch.zeros(8, 1); q.zero_point = torch.full((8, 1), -16.6); q.calibrated = True
    assert q.qparams_for(256)[0].numel() == 1
    q.scale = torch.arange(8.0).reshape(8, 1); q._epoch += 1
    with pytest.raises(RuntimeError, match="does not fit"):
        q.qparams_for(256)
    assert q.qparams_for(8)[0].numel() == 8            # a genuine per-channel vector of the expected length passes through
