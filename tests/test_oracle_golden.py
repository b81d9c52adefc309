"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors produced by the reference
(tests/golden/make_golden.py).  Elementwise quantities must be bit-identical."""
import hashlib
import json
import os

import pytest
import torch

from oracle import ref_cpu as O
from helpers import GOLDEN, LAYER_CASES, QUANT_CASES, assert_close_y, load_case


def _beq(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.equal(a, b), f"{what}: not bit-identical"


@pytest.mark.parametrize("name", LAYER_CASES)
def test_layer_case(name):
    meta, t = load_case(name)
    ol = O.build_calibrated_layer(t["W"], t["bias"], t["A"], t["B"], [t["x0"], t["x1"]], meta["bits"],
                                  meta["qtype"], meta["per_channel"], meta["alpha"], meta["r"])
    for tag, q in (("qx", ol.qx), ("qw", ol.qw), ("qA", ol.qA), ("qB", ol.qB)):
        for k in ("scale", "zero_point", "running_min", "running_max"):
            _beq(getattr(q, k), t[f"{tag}.{k}"], f"{name}.{tag}.{k}")
    _beq(ol.qx(t["x2"]), t["fq_x2"], "fq_x2")
    _beq(ol.qx(t["x0"]), t["fq_x0"], "fq_x0")
    _beq(ol.qw(t["W"]), t["fq_W"], "fq_W")
    _beq(ol.qA(t["A"]), t["fq_A"], "fq_A")
    _beq(ol.qB(t["B"]), t["fq_B"], "fq_B")
    for tag, q, src in (("lv_x2", ol.qx, "x2"), ("lv_W", ol.qw, "W"), ("lv_A", ol.qA, "A"), ("lv_B", ol.qB, "B")):
        _beq(q.levels(t[src]).to(torch.int32), t[tag], tag)
    # GEMM outputs: summation order of ATen sgemm may differ across hosts -> tolerance
    assert_close_y(ol.forward(t["x2"], calibration_mode=True), t["base_x2"], "base_x2")
    assert_close_y(ol.forward(t["x2"]), t["y_x2"], "y_x2")
    assert_close_y(ol.forward(t["x0"]), t["y_x0"], "y_x0")
    assert_close_y(ol.forward(t["x2"].reshape(-1, meta["K"])[:40]), t["y_2d"], "y_2d")


@pytest.mark.parametrize("name", QUANT_CASES)
def test_quantizer_case(name):
    meta, t = load_case(name)
    q = O.QuantState(meta["bits"], meta["qtype"], meta["channel_dim"], meta["per_channel"], meta["symmetric"])
    q.start()
    for i in range(meta["batches"]):
        q(t[f"x{i}"])
    q.finish()
    for k in ("scale", "zero_point", "running_min", "running_max"):
        _beq(getattr(q, k), t[k], f"{name}.{k}")
    _beq(q(t["xt"]), t["fq_xt"], "fq_xt")
    _beq(q.levels(t["xt"]).to(torch.int32), t["lv_xt"], "lv_xt")


def test_config1_full_weight_per_tensor_8bit():
    """BASELINE.json configs[0]: LearnableFakeQuantize minmax 8-bit per-tensor on the 768x3072 weight."""
    js = json.load(open(os.path.join(GOLDEN, "config1_checksums.json")))
    W = O.make_workload(8, 768, 3072, 64, seed=0)[0]
    assert hashlib.sha256(W.numpy().tobytes()).hexdigest() == js["W_sha256"], "synthetic W generator drifted"
    q = O.QuantState(8, "minmax", 0, False).calibrate_on(W)
    assert list(q.scale.shape) == js["scale_shape"] == [1, 1]
    assert q.scale.flatten()[0].item().hex() == js["scale_hex"]
    lv = q.levels(W).to(torch.int64)
    hist = torch.bincount((lv + 127).flatten(), minlength=255).tolist()
    assert hist == js["level_hist_from_-127"]
    assert int((torch.tensor(hist) > 0).sum()) == js["distinct_levels"] == 239
    assert (int(lv.min()), int(lv.max())) == (js["level_min"], js["level_max"])
    assert int(lv.sum()) == js["level_sum"] and int(lv.abs().sum()) == js["level_abs_sum"]
    fq = q(W)
    assert hashlib.sha256(fq.numpy().tobytes()).hexdigest() == js["fq_sha256"]
    assert [v.hex() for v in fq.flatten()[:64].tolist()] == js["fq_first64_hex"]


def test_uncalibrated_raises():
    q = O.QuantState(8)
    with pytest.raises(RuntimeError):
        q(torch.ones(3))


GRAD_CASES = ["mm4_pc", "log6_pc", "mm8_pt"]


def load_grad_case(name):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, f"grad_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


@pytest.mark.parametrize("name", GRAD_CASES)
def test_backward_closed_form_matches_reference_autograd(name):
    """The straight-through backward (quantization_methods.py:25-28, 82-90) restated in closed form against the grads
    the reference's autograd produced (fixtures from make_golden.py)."""
    meta, t = load_grad_case(name)
    ol = O.build_calibrated_layer(t["W"], t["bias"], t["A"], t["B"], [t["x0"], t["x1"]], meta["bits"], meta["qtype"],
                                  meta["per_channel"], meta["alpha"], meta["r"])
    assert_close_y(ol.forward(t["xg"]), t["y"], "y")
    gx, gA, gB = O.sp_linear_backward(ol, t["xg"], t["g"])
    for got, want, what in ((gx, t["grad_x"], "grad_x"), (gA, t["grad_A"], "grad_A"), (gB, t["grad_B"], "grad_B")):
        assert_close_y(got, want, f"{name}.{what}", 1e-5)
