"""NaN handling like the reference's torch ops: `tensor.min/max`, `torch.minimum/maximum` and `torch.clamp` propagate
NaN, so calibrating on a diverged activation yields a NaN scale for that channel (not a plausible finite one), a NaN element
fake-quantizes to NaN, and a NaN activation makes its output row NaN."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


@pytest.mark.parametrize("qtype", ["minmax", "log"])
@pytest.mark.parametrize("per_channel", [True, False])
def test_statistics_propagate_nan(pkg, qtype, per_channel):
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(4, 96, 80, generator=g) for _ in range(3)]
    xs[1][2, 17, 11] = float("nan")                     # second batch, one element
    for shape_case in ("input", "weight"):
        if shape_case == "input":
            q = pkg.LearnableFakeQuantize(6, channel_dim=-1, quantizer_type=qtype, per_channel=per_channel, is_input=True)
            oq = O.QuantState(6, qtype, -1, per_channel)
            batches = xs
        else:
            q = pkg.LearnableFakeQuantize(6, channel_dim=0, quantizer_type=qtype, per_channel=per_channel)
            oq = O.QuantState(6, qtype, 0, per_channel)
            batches = [x.reshape(-1, 80)[:128].t().contiguous() for x in xs]      # [80, 128], NaN lands in row 11
            batches[1][11, 5] = float("nan")
        q.start_calibration(); oq.start()
        for b in batches:
            q(b.to(DEV)); oq.observe(b)
        q.finish_calibration(); oq.finish()
        for name, ref in (("running_min", oq.running_min), ("running_max", oq.running_max), ("scale", oq.scale),
                          ("zero_point", oq.zero_point)):
            got = getattr(q, name).cpu()
            assert got.shape == ref.shape, (shape_case, name)
            assert torch.equal(torch.isnan(got), torch.isnan(ref)), f"{shape_case}.{name}: NaN pattern differs from torch's"
            ok = ~torch.isnan(ref)
            if qtype == "minmax":
                assert torch.equal(got[ok], ref[ok]), (shape_case, name)
            else:
                assert torch.allclose(got[ok], ref[ok], rtol=1e-6, atol=1e-6), (shape_case, name)
        assert bool(torch.isnan(q.scale).any())
        if per_channel:
            assert int(torch.isnan(q.scale).sum()) == 1         # only the channel that saw the NaN


@pytest.mark.parametrize("qtype", ["minmax", "log"])
def test_fakequant_keeps_nan_elements(pkg, qtype):
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(4)
    x = torch.randn(64, 48, generator=g)
    oq = O.QuantState(5, qtype, 0, True).calibrate_on(x)
    q = pkg.LearnableFakeQuantize(5, channel_dim=0, quantizer_type=qtype)
    q.start_calibration(); q(x.to(DEV)); q.finish_calibration()
    xn = x.clone(); xn[7, 3] = float("nan"); xn[20, 40] = float("nan")
    got, ref = q(xn.to(DEV)).cpu(), oq(xn)
    assert torch.equal(torch.isnan(got), torch.isnan(ref)) and int(torch.isnan(got).sum()) == 2
    ok = ~torch.isnan(ref)
    assert torch.allclose(got[ok], ref[ok], rtol=4e-6, atol=0) if qtype == "log" else torch.equal(got[ok], ref[ok])


@pytest.mark.parametrize("bits,qtype", [(4, "minmax"), (6, "log")])
def test_fused_forward_nan_row(pkg, bits, qtype):
    from oracle import ref_cpu as O
    M, K, N, r = 256, 128, 192, 16
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=2, batch=2)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: r, 32: 0}, {bits: qtype, 32: None})
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    x = x0.clone().to(DEV)
    with torch.no_grad():
        y = layer(x)
        x[1, 5, 9] = float("nan")
        yn = layer(x)
    nan_rows = torch.isnan(yn).all(dim=-1)
    assert int(nan_rows.sum()) == 1 and bool(nan_rows[1, 5])            # the whole row of that token, as F.linear gives
    assert torch.equal(yn[~nan_rows], y[~nan_rows])


@pytest.mark.parametrize("path", ["I8", "AUTO"])
@pytest.mark.parametrize("lora_on", [True, False])
def test_fused_forward_nan_row_byte_level_paths(pkg, path, lora_on):
    """the byte-level operand path (int8 levels; per-tensor input scale: what PATH_AUTO picks for every evaluation-loader model)
    cannot hold a NaN level.  With the LoRA branch the fp32 LoRA-down product carries the NaN; without it (calibration_mode,
    disabled adapter) the layer must not take a byte-level path: either way the token's whole output row is NaN, as F.linear's."""
    from oracle import ref_cpu as O
    M, K, N, r = 256, 128, 192, 16
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=5, batch=2)
    layer = pkg.SPLinearWithLoRA(K, N, [8, 32], {8: r, 32: 0}, {8: r, 32: 0}, {8: "minmax", 32: None}, per_channel=False)
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters["8bit"].lora_A.copy_(A); layer.lora_adapters["8bit"].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(8)
    pkg.calibrate_layer(layer, 8, [x0.to(DEV), x1.to(DEV)])
    layer.operand_path = {"I8": pkg._lib.PATH_I8, "AUTO": pkg._lib.PATH_AUTO}[path]
    if not lora_on:
        layer.lora_adapters["8bit"].enabled = False
    x = x0.clone().to(DEV)
    with torch.no_grad():
        y = layer(x)
        x[1, 5, 9] = float("nan")
        yn = layer(x)
    nan_rows = torch.isnan(yn).all(dim=-1)
    assert int(nan_rows.sum()) == 1 and bool(nan_rows[1, 5])
    assert not bool(torch.isnan(yn[~nan_rows]).any())
    assert torch.equal(yn[~nan_rows], y[~nan_rows])
