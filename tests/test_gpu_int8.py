"""GPU parity of the int8 operand path (SPQ_PATH_I8: int8 activation levels x the weight's own int8 levels on
v_mfma_i32_32x32x32_i8, exact i32 sums) against the fixtures the reference produced and, on further shapes, against the oracle.
Valid when the input scale is per tensor and the weights are symmetric minmax of <= 8 bits -- the configuration the reference's
evaluation loader forces (deploy.py:210,238); every other layer asked for the path takes the fp16-limb path instead.
Same bar as the other operand paths: |d| <= 1e-5 |y_ref| + 1e-5 rms(y_ref)."""
import pytest
import torch

from helpers import LAYER_CASES, assert_close_y, load_case
from test_gpu_parity import build_layer, DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


def _pin_scales(layer, key, t):
    lora = layer.lora_adapters[key]
    quants = {"qx": layer.quantizers_input[key], "qw": layer.quantizers_weight[key], "qA": lora.quantize_A, "qB": lora.quantize_B}
    with torch.no_grad():
        for tag, q in quants.items():
            if tuple(q.scale.shape) == tuple(t[f"{tag}.scale"].shape):
                q.scale = t[f"{tag}.scale"].to(DEV); q.zero_point = t[f"{tag}.zero_point"].to(DEV); q._epoch += 1


@pytest.mark.parametrize("name", [n for n in LAYER_CASES if n.startswith("mm")])
def test_layer_case_int8(pkg, name):
    meta, t = load_case(name)
    L = pkg._lib
    if meta["bits"] > 12:
        want = L.PATH_F16X3
    elif meta["bits"] <= 8 and not meta["per_channel"]:
        want = L.PATH_I8
    else:
        want = L.PATH_F16X2                                # per-channel input scale (or > 8 bit): the scale cannot leave the sum
    layer, key = build_layer(pkg, meta, t, L.PATH_I8)
    _pin_scales(layer, key, t)
    with torch.no_grad():
        y2 = layer(t["x2"].to(DEV))
        assert layer._last_path == want, (layer._last_path, want)
        y0 = layer(t["x0"].to(DEV))
        layer.calibration_mode = True
        base = layer(t["x2"].to(DEV))
        layer.calibration_mode = False
        y2d = layer(t["x2"].reshape(-1, meta["K"])[:40].contiguous().to(DEV))
        # re-quantising forward (with and without the preparation inside the activation pass) == cached operands
        layer.cache_operands = False
        for fuse in (False, True):
            layer.fuse_prepare = fuse
            assert torch.equal(layer(t["x2"].to(DEV)), y2)
    assert_close_y(y2, t["y_x2"], f"{name}.y_x2[i8]", 1e-5)
    assert_close_y(y0, t["y_x0"], f"{name}.y_x0[i8]", 1e-5)
    assert_close_y(base, t["base_x2"], f"{name}.base_x2[i8]", 1e-5)
    assert_close_y(y2d, t["y_2d"], f"{name}.y_2d[i8]", 1e-5)


SHAPES = [  # M, K, N, r, bits
    (4096, 768, 3072, 64, 8),        # BASELINE config 2, per-tensor variant: 128-deep stages
    (8192, 768, 3072, 64, 4),        # the headline shape with a per-tensor scale
    (1000, 3072, 768, 64, 4),        # mlp.c_proj: K = 3072, ragged M
    (300, 64, 130, 0, 8),            # no LoRA (-> fp16 levels, see below), N % 4 != 0, K = 64
    (300, 64, 132, 8, 8),            # K = 64: the 64-deep ring kernel
    (512, 192, 1024, 100, 6),        # rank 100, K % 128 != 0: ring kernel
    (256, 1024, 256, 16, 2),         # 2-bit
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_int8_path_against_oracle(pkg, shape):
    from oracle import ref_cpu as O
    M, K, N, r, bits = shape
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, max(r, 1), seed=M + bits, batch=1)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: max(r, 1), 32: 0}, {bits: "minmax", 32: None},
                                 per_channel=False)
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        if r:
            layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval()
    layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    if r:
        y_ref = O.build_calibrated_layer(W, bias, A, B, [x0, x1], bits, "minmax", False, r, r).forward(x1)
    else:
        qw = O.QuantState(bits, "minmax", 0, False).calibrate_on(W)
        qx = O.QuantState(bits, "minmax", -1, False)
        qx.start(); qx.observe(x0); qx.observe(x1); qx.finish()
        y_ref = O.sp_linear_forward(x1, W, bias, qx, qw, bits=bits)
    outs = {}
    with torch.no_grad():
        for pname, pth in (("auto", pkg._lib.PATH_AUTO), ("f16x2", pkg._lib.PATH_F16X2), ("i8", pkg._lib.PATH_I8)):
            layer.operand_path = pth
            outs[pname] = layer(x1.to(DEV))
            if pname != "f16x2":
                # AUTO picks the int8 path wherever it is valid -- with a LoRA term (without one a NaN activation would go unseen
                # in the byte-level operand: such forwards take the fp16-level path, tests/test_gpu_nan.py)
                assert layer._last_path == (pkg._lib.PATH_I8 if r else pkg._lib.PATH_F16X2)
            assert_close_y(outs[pname], y_ref, f"{pname} {shape}", 1e-5)
        assert torch.equal(outs["auto"], outs["i8"])
        y_gelu = layer(x1.to(DEV), activation="gelu")               # the GELU epilogue of the int8 kernels
        # (fp16-level paths fuse the GELU when rows of y are 16-B aligned; the ragged-N instantiation has the plain epilogue only)
        assert layer._activation_fused == (layer._last_path == pkg._lib.PATH_I8 or N % 4 == 0)
    assert_close_y(y_gelu, torch.nn.functional.gelu(y_ref), f"gelu {shape}", 1e-5)
