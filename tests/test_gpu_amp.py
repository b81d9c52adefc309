"""The AMP contract (SURVEY.md §8 f2 "AMP interaction must be defined").  The reference trains under
``torch.amp.autocast('cuda')`` (train_sp.py:319, part2's train_cpt.py): a producer may hand the layer a half tensor.  This build's
kernels are fp32: a half input is promoted exactly (every fp16 / bf16 value is an fp32 value), the output is fp32, and the
straight-through gradient comes back in the input's dtype.  So under autocast the layer must give bit for bit what it gives, outside
autocast, for the promoted input -- and that result must meet the oracle's bar -- forward and backward, part1 and part2."""
import pytest
import torch

from helpers import assert_close_y

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


def _sp_layer(pkg, bits, qtype, per_channel, train):
    from oracle import ref_cpu as O
    M, K, N, r = 512, 256, 320, 16
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=11, batch=2)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: r, 32: 0}, {bits: qtype, 32: None}, per_channel=per_channel)
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV)
    layer.set_precision(bits)
    pkg.calibrate_layer(layer.eval(), bits, [x0.to(DEV), x1.to(DEV)])
    layer.linear.weight.requires_grad_(False)                      # main_sp.py:83: the base weight is frozen during QAT
    oracle = O.build_calibrated_layer(W, bias, A, B, [x0, x1], bits, qtype, per_channel, r, r)
    return (layer.train() if train else layer.eval()), oracle, x1


@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bits,qtype,per_channel", [(4, "minmax", True), (8, "minmax", False), (6, "log", True)])
def test_sp_linear_eval_under_autocast(pkg, bits, qtype, per_channel, half):
    layer, oracle, x1 = _sp_layer(pkg, bits, qtype, per_channel, train=False)
    xh = x1.to(DEV).to(half)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=half):
            y_amp = layer(xh)
        y_f32 = layer(xh.float())
    assert y_amp.dtype == torch.float32 and y_amp.shape == (*x1.shape[:-1], layer.out_features)
    assert torch.equal(y_amp, y_f32), "autocast changed the result for the same (promoted) input"
    assert_close_y(y_amp, oracle.forward(xh.float().cpu()), f"amp eval {bits}-bit {qtype} {half}", 1e-5)


@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bits,qtype,per_channel", [(4, "minmax", True), (6, "log", True)])
def test_sp_linear_train_step_under_autocast(pkg, bits, qtype, per_channel, half):
    layer, oracle, x1 = _sp_layer(pkg, bits, qtype, per_channel, train=True)
    key = f"{bits}bit"
    lora = layer.lora_adapters[key]
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(*x1.shape[:-1], layer.out_features, generator=g).to(DEV)

    def step(x, amp):
        for p_ in layer.parameters():
            p_.grad = None
        x = x.detach().clone().requires_grad_(True)
        if amp:
            with torch.autocast("cuda", dtype=half):
                y = layer(x)
        else:
            y = layer(x)
        assert y.dtype == torch.float32
        y.backward(gy)
        return y.detach(), x.grad, lora.lora_A.grad.clone(), lora.lora_B.grad.clone()

    xh = x1.to(DEV).to(half)
    y_a, gx_a, gA_a, gB_a = step(xh, True)
    y_f, gx_f, gA_f, gB_f = step(xh.float(), False)
    assert gx_a.dtype == half and gx_f.dtype == torch.float32                     # the gradient comes back in the input's dtype
    assert gA_a.dtype == torch.float32 and gB_a.dtype == torch.float32
    assert torch.equal(y_a, y_f) and torch.equal(gA_a, gA_f) and torch.equal(gB_a, gB_f)
    assert torch.equal(gx_a, gx_f.to(half)), "d/dx under autocast is not the fp32 gradient rounded once to the input's dtype"
    assert_close_y(y_a, oracle.forward(xh.float().cpu()), f"amp train fwd {bits}-bit {qtype} {half}", 1e-5)
    assert bool(torch.isfinite(gx_a.float()).all()) and float(gA_a.abs().max()) > 0 and float(gB_a.abs().max()) > 0


@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
def test_cpt_linear_under_autocast(pkg, half):
    from test_cpt_cpu import load_cpt
    from test_gpu_cpt import build
    meta, t = load_cpt("minmax_4_8")
    student = [b for b in meta["widths"] if b < 32 and b not in meta["skip_calibration"]]
    b = student[0]
    m = build(pkg, meta, t, train=False)
    x0, x1, x2 = t["x0"].to(DEV), t["x1"].to(DEV), t["x2"].to(DEV)
    pkg.calibrate_cpt_layer(m, b, [x0, x1])
    m.set_precision(b)
    xh = x2.to(half)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=half):
            y_amp = m(xh)
        y_f32 = m(xh.float())
    assert y_amp.dtype == torch.float32 and torch.equal(y_amp, y_f32)
    # training step: gradients of the shared LoRA factors and of the input
    m.train()
    m.linear.weight.requires_grad_(False)
    gy = torch.randn_like(y_f32)

    def step(x, amp):
        for p_ in m.parameters():
            p_.grad = None
        x = x.detach().clone().requires_grad_(True)
        if amp:
            with torch.autocast("cuda", dtype=half):
                y = m(x)
        else:
            y = m(x)
        y.backward(gy)
        return y.detach(), x.grad, m.shared_lora.lora_A.grad.clone(), m.shared_lora.lora_B.grad.clone()

    y_a, gx_a, gA_a, gB_a = step(xh, True)
    y_f, gx_f, gA_f, gB_f = step(xh.float(), False)
    assert y_a.dtype == torch.float32 and gx_a.dtype == half
    assert torch.equal(y_a, y_f) and torch.equal(gA_a, gA_f) and torch.equal(gB_a, gB_f) and torch.equal(gx_a, gx_f.to(half))
