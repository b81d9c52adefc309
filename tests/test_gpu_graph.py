"""GPU: a forward captured into a HIP graph (llm_qat_on_gpt2_amd.GraphedForward) replays bit for bit what the eager call computes:
one layer (re-quantising and with cached operands), a block stack, and part2's fused feed-forward; a re-quantising capture sees
new LoRA weights, a wrong input shape is refused."""
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_on_gpt2_amd as p
    p._lib.load()
    return p


def make_layer(pkg, K=256, N=384, r=32, bits=4, M=512, seed=0):
    from llm_qat_on_gpt2_amd import synthetic as S
    W, bias, A, B, x0, x1 = S.make_workload(M, K, N, r, seed=seed, batch=4)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: r, 32: 0}, {bits: "minmax", 32: None})
    key = f"{bits}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(DEV).eval(); layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(DEV), x1.to(DEV)])
    return layer, x0.to(DEV), x1.to(DEV)


@pytest.mark.parametrize("cache", [True, False])
def test_layer_replay_is_bit_identical(pkg, cache):
    layer, x0, x1 = make_layer(pkg)
    layer.cache_operands = cache
    with torch.no_grad():
        want0, want1 = layer(x0).clone(), layer(x1).clone()
    g = pkg.GraphedForward(layer, x0)
    assert torch.equal(g(x0), want0)
    assert torch.equal(g(x1), want1)                       # new input values through the captured buffers
    assert torch.equal(g(x0), want0)
    with pytest.raises(RuntimeError):
        g(x0[:, :5])                                         # another shape needs another capture


def test_requantising_capture_sees_new_lora_weights(pkg):
    layer, x0, _ = make_layer(pkg, seed=3)
    layer.cache_operands = False                             # the graph holds the preparation launch: it reads the live weights
    g = pkg.GraphedForward(layer, x0)
    before = g(x0).clone()
    with torch.no_grad():
        layer.lora_adapters["4bit"].lora_B.mul_(0.5)         # in place: same pointer, new values (still inside the calibrated range)
        want = layer(x0).clone()
    got = g(x0)
    assert not torch.equal(before, want)
    assert torch.equal(got, want)


def test_block_stack_and_cpt_pair_replay(pkg):
    E, bits = 128, 4
    cfg = types.SimpleNamespace(n_embd=E, n_head=4, n_positions=64, layer_norm_epsilon=1e-5, bit_widths=[bits, 32],
                                lora_rank_per_bit={bits: 16, 32: 0}, lora_alpha_per_bit={bits: 16, 32: 0},
                                quantizer_per_bit={bits: "minmax", 32: None}, per_channel_quantization=True)
    torch.manual_seed(0)
    blocks = torch.nn.ModuleList([pkg.SPBlock(cfg, bit_widths=[bits, 32]) for _ in range(2)])
    with torch.no_grad():
        for n, p in blocks.named_parameters():
            if "lora_B" in n: p.normal_(0, 0.01)
            elif p.dim() > 1 and "lora_A" not in n: p.normal_(0, 0.02)
    blocks = blocks.to(DEV).eval()

    def stack(x):
        for b in blocks: x = b(x)
        return x
    model = types.SimpleNamespace(modules=blocks.modules, named_modules=blocks.named_modules, __call__=stack)
    xs = [torch.randn(2, 64, E, device=DEV) for _ in range(3)]
    for b in blocks: b.set_precision(bits)

    class M(torch.nn.Module):
        def __init__(self): super().__init__(); self.h = blocks
        def forward(self, x): return stack(x)
    pkg.calibrate_model(M().to(DEV).eval(), bits, xs[:2])
    with torch.no_grad():
        want = stack(xs[2]).clone()
    g = pkg.GraphedForward(stack, xs[0])
    assert torch.equal(g(xs[2]), want)

    # part2's fused feed-forward (levels-out store) under replay
    from test_gpu_cpt_mlp import make_chain
    fc_in, fc_out, _, _, x = make_chain(pkg, 128, 512, 4, "minmax", 256, seed=11)
    xd = x.to(DEV)
    with torch.no_grad():
        want = pkg.cpt_mlp_forward(fc_in, fc_out, xd).clone()
    g2 = pkg.GraphedForward(lambda t: pkg.cpt_mlp_forward(fc_in, fc_out, t), xd)
    assert torch.equal(g2(xd), want)
