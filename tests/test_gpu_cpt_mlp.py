"""GPU: CPTBlock's feed-forward fc_out(gelu(fc_in(x))) (cpt_model.py:196-198) with the levels-out store (second
half): fc_in's contraction writes fc_out's input LEVELS, the fp32 activation is never stored.  Checked (i) level for level
against the quantizer applied to the fp32 activation the same kernel stores, (ii) bit for bit against the two-launch chain on
that activation, (iii) against the oracle's chain at the usual bound with the level-flip allowance of a chained test."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close_y

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


def make_chain(pkg, E, H, bits, qtype, M, seed, rank=16):
    from oracle import ref_cpt as O
    g = torch.Generator().manual_seed(seed)
    widths = [bits, 32]
    qpb = {bits: qtype, 32: None}
    t = {}
    layers, oracles = [], []
    for K, N in ((E, H), (H, E)):
        W = torch.randn(N, K, generator=g) * 0.02
        bias = torch.randn(N, generator=g) * 0.02
        A = (torch.rand(K, rank, generator=g) - 0.5) * (2.0 / K ** 0.5)
        B = torch.randn(N, rank, generator=g) * 0.01
        m = pkg.CPTLinear(K, N, bit_widths=widths, quantizer_per_bit=qpb, shared_lora_rank=rank, shared_lora_alpha=32)
        with torch.no_grad():
            m.linear.weight.copy_(W); m.linear.bias.copy_(bias)
            m.shared_lora.lora_A.copy_(A); m.shared_lora.lora_B.copy_(B)
        layers.append(m.to(DEV).eval())
        oracles.append(O.OracleCPTLayer(W, bias, A, B, widths, qpb, rank=rank, alpha=32.0))
    xs = [torch.randn(4, M // 4, E, generator=g) for _ in range(3)]
    fc_in, fc_out = layers
    o_in, o_out = oracles
    # calibration: fc_in on x, fc_out on gelu(fc_in(x)) of the calibrated fc_in (product and oracle each on their own chain)
    pkg.calibrate_cpt_layer(fc_in, bits, [x.to(DEV) for x in xs[:2]])
    o_in.calibrate(bits, xs[:2])
    with torch.no_grad():
        hs = [F.gelu(fc_in(x.to(DEV))) for x in xs[:2]]
    pkg.calibrate_cpt_layer(fc_out, bits, hs)
    o_out.calibrate(bits, [F.gelu(o_in.forward(x)) for x in xs[:2]])
    if qtype == "log":
        # log-domain statistics agree with the oracle's to 1 ulp only (log path), and fc_out's were taken on the
        # product's own activations: pin every scale to the oracle's, as tests/test_gpu_cpt.py does with the golden values
        for m, o in ((fc_in, o_in), (fc_out, o_out)):
            for q, oq in ((m.quantizer_input, o.q_in), (m.quantizer_weight, o.q_w), (m.lora_weight_quantizers[f"{bits}bit"], o.q_lora[bits])):
                assert tuple(q.scales[bits].shape) == tuple(oq.scales[bits].shape)
                q.scales[bits] = oq.scales[bits].to(DEV)
                q.zero_points[bits] = oq.zero_points[bits].to(DEV)
                q._epoch += 1
    return fc_in, fc_out, o_in, o_out, xs[2]


@pytest.mark.parametrize("E,H,bits,qtype,M", [(128, 512, 4, "minmax", 512), (192, 768, 8, "minmax", 1000), (768, 3072, 4, "minmax", 2048),
                                              (128, 512, 6, "log", 512), (256, 1024, 4, "log", 1024)])
def test_cpt_mlp_levels_out(pkg, E, H, bits, qtype, M):
    if M % 4:
        M -= M % 4
    fc_in, fc_out, o_in, o_out, x = make_chain(pkg, E, H, bits, qtype, M, seed=E + bits)
    xd = x.to(DEV)
    with torch.no_grad():
        y_two = fc_out(F.gelu(fc_in(xd)))                        # plain: two layers, stock gelu
        y_fused = pkg.cpt_mlp_forward(fc_in, fc_out, xd)
        # (i) + (ii): the activation the fused store WOULD have written in fp32 (same kernel, GELU in the store) ...
        x2 = xd.view(-1, E)
        h = fc_in._gemm.run(x2, fc_in.linear.bias, fc_in.quantizer_input, True, epilogue=pkg._lib.EPILOGUE_GELU)
        qi2 = fc_out.quantizer_input
        Mtot = x2.shape[0]
        Mp = (Mtot + 255) // 256 * 256
        buf = pkg._lib.workspace(xd.device, 1, slot=1)          # the stream's level-matrix slot, as the fused call left it
        plane = lambda i: buf[i * Mp * H * 2: (i + 1) * Mp * H * 2].view(torch.float16).view(-1, H)[:Mtot]
        if qtype == "minmax":                                    # integer levels: the quantizer's own level kernel
            lv_ref = qi2.quantize_levels(h)
            lv = plane(0)
            assert torch.equal(lv.to(torch.int32).reshape(-1), lv_ref.to(torch.int32).reshape(-1)), "levels-out differ from the input quantizer's levels"
            assert int(lv.abs().max()) <= (1 << (bits - 1)) - 1
        else:                                                    # two fp16 limbs of FQ(h) * 2^G: hi + lo is FQ(h) * 2^G to 2^-22
            from llm_qat_on_gpt2_amd.sp_linear import _limb_scale
            p2 = _limb_scale(qi2)
            fq = qi2(h).reshape(Mtot, H).double() * p2[0].double()
            got = plane(0).double() + plane(1).double()
            assert bool(((got - fq).abs() <= fq.abs() * 2.0 ** -21 + 1e-30).all()), "limbs-out do not add up to FQ(h) * 2^G"
        y_chain = fc_out(h.view(*xd.shape[:-1], H))              # ... and the ordinary second layer on it
        assert torch.equal(y_fused, y_chain), "fused pair differs from the two-launch chain on the same activation"
    # (iii) the oracle's chain; an h within rounding distance of a level boundary may flip a level of fc_out's input
    ref = o_out.forward(F.gelu(o_in.forward(x)))
    tol = 1e-5
    bound = tol * ref.abs() + tol * ref.pow(2).mean().sqrt()
    for what, y in (("fused pair", y_fused), ("two layers, stock gelu", y_two)):
        bad_rows = ((y.cpu() - ref).abs() > bound).any(dim=-1).float().mean().item()
        assert bad_rows <= 0.02, f"{what}: {bad_rows:.3%} of rows beyond the bound"


def test_cpt_mlp_falls_back_under_autograd(pkg):
    fc_in, fc_out, o_in, o_out, x = make_chain(pkg, 128, 512, 4, "minmax", 256, seed=3)
    xd = x.to(DEV).requires_grad_(True)
    y = pkg.cpt_mlp_forward(fc_in, fc_out, xd)                   # grad mode on: the ordinary autograd path
    y.sum().backward()
    assert xd.grad is not None and torch.isfinite(xd.grad).all()


def test_levels_out_rejected_where_unsupported(pkg):
    """the C ABI refuses the levels-out store on a shape it cannot serve (N % 64 != 0)"""
    lib = pkg._lib.load()
    fc_in, fc_out, *_ = make_chain(pkg, 128, 512, 4, "minmax", 256, seed=5)
    x2 = torch.randn(64, 128, device=DEV)
    with torch.no_grad():
        fc_in(x2)                                                # operands prepared
    buf = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    q = fc_out.quantizer_input
    with pytest.raises(pkg._lib.SpqError):
        fc_in._gemm.run(x2, fc_in.linear.bias, fc_in.quantizer_input, True, levels_out=(buf, 500, q))   # row pitch < N


@pytest.mark.parametrize("seed", range(8))
def test_cpt_mlp_random_shapes_match_the_two_launch_chain(pkg, seed):
    """random hidden sizes (multiples of 64), token counts (ragged against the 128 / 256-row tiles), widths and quantizer types:
    the fused pair is the two-launch chain on the same activation, bit for bit"""
    g = torch.Generator().manual_seed(9000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    E = 8 * ri(4, 40)
    H = 64 * ri(1, 12)
    M = 4 * ri(1, 300)
    qtype = "minmax" if seed % 2 == 0 else "log"
    bits = ri(2, 12) if qtype == "minmax" else ri(2, 8)
    fc_in, fc_out, _, _, x = make_chain(pkg, E, H, bits, qtype, M, seed=seed, rank=ri(1, 32))
    xd = x.to(DEV)
    with torch.no_grad():
        y_fused = pkg.cpt_mlp_forward(fc_in, fc_out, xd)
        h = fc_in._gemm.run(xd.view(-1, E), fc_in.linear.bias, fc_in.quantizer_input, True, epilogue=pkg._lib.EPILOGUE_GELU)
        y_chain = fc_out(h.view(*xd.shape[:-1], H))
        y_plain = fc_out(F.gelu(fc_in(xd)))
    assert fc_out._last_path in (pkg._lib.PATH_F16X2, pkg._lib.PATH_F16X3)
    assert torch.equal(y_fused, y_chain), (E, H, M, bits, qtype)
    # against the plain three calls: the same function up to level flips of single elements of h
    close = ((y_fused - y_plain).abs() <= 1e-5 * y_plain.abs() + 1e-5 * y_plain.pow(2).mean().sqrt()).all(dim=-1).float().mean().item()
    assert close >= 0.97, (close, E, H, M, bits, qtype)
