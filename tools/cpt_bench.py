"""Forward (and forward+backward) time of part2's CPTLinear on the HIP path at the c_fc shape, next to part1's layer."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as C          # seeded input generator
dev = 'cuda:0'
PATHN = {1: 'f32', 2: 'f16x2', 3: 'u8x2', 4: 'f16x3', 5: 'i8'}


def timeit(fn, n=50):
    """best of three runs of n calls: a run that meets an allocator event (a fresh hipMalloc behind torch's caching allocator)
    reads several times too long"""
    for _ in range(5): fn()
    best = float('inf')
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


for (M, K, N, r, bits, qt) in [(8192, 768, 3072, 16, 4, 'minmax'), (8192, 768, 3072, 16, 6, 'log'), (8192, 3072, 768, 16, 6, 'log'),
                               (8192, 768, 3072, 16, 8, 'minmax')]:
    W, bias, A, B, x0, x1 = C.make_cpt_workload(M, K, N, r, seed=0, batch=8)
    m = pkg.CPTLinear(K, N, bit_widths=[bits, 32], quantizer_per_bit={bits: qt, 32: None}, shared_lora_rank=r, shared_lora_alpha=32)
    with torch.no_grad():
        m.linear.weight.copy_(W); m.linear.bias.copy_(bias); m.shared_lora.lora_A.copy_(A); m.shared_lora.lora_B.copy_(B)
    m = m.to(dev).eval()
    pkg.calibrate_cpt_layer(m, bits, [x0.to(dev), x1.to(dev)])
    m.set_precision(bits)
    x = x0.to(dev)
    flop = 2 * M * (K * N + K * r + r * N)
    with torch.no_grad():
        t_eval = timeit(lambda: m(x))
        m.cache_operands = False
        t_requant = timeit(lambda: m(x))
        m.cache_operands = True
    m.train()
    m.linear.weight.requires_grad_(False); m.linear.bias.requires_grad_(False)
    xg = x.clone().requires_grad_(True)
    g = torch.randn_like(m(xg)) * 1e-3

    def step():
        xg.grad = None
        m(xg).backward(g)
    t_train = timeit(step, 30)
    print(f'CPT {qt}{bits} M={M} K={K} N={N} r={r} path={PATHN[m._last_path]}: eval {t_eval*1e3:.4f} ms ({flop/t_eval/1e12:.0f} TFLOP/s), '
          f'weights re-quantized every call {t_requant*1e3:.4f} ms ({flop/t_requant/1e12:.0f} TFLOP/s), fwd+bwd {t_train*1e3:.3f} ms', flush=True)


# CPTBlock's feed-forward fc_out(gelu(fc_in(x))) (cpt_model.py:196-198): two layers + stock gelu against the fused pair
# (GELU in fc_in's store, fc_out's input levels written by that store, no fp32 activation: cpt_mlp_forward)
import torch.nn.functional as F
for (M, E, bits, qt) in [(8192, 768, 4, 'minmax'), (32768, 768, 4, 'minmax'), (8192, 768, 8, 'minmax'), (8192, 1024, 4, 'minmax'),
                         (8192, 768, 6, 'log'), (32768, 768, 6, 'log'), (8192, 768, 4, 'log')]:
    H, r = 4 * E, 16
    layers = []
    for (K, N) in ((E, H), (H, E)):
        W, bias, A, B, x0, x1 = C.make_cpt_workload(256, K, N, r, seed=1, batch=4)
        m = pkg.CPTLinear(K, N, bit_widths=[bits, 32], quantizer_per_bit={bits: qt, 32: None}, shared_lora_rank=r, shared_lora_alpha=32)
        with torch.no_grad():
            m.linear.weight.copy_(W); m.linear.bias.copy_(bias); m.shared_lora.lora_A.copy_(A); m.shared_lora.lora_B.copy_(B)
        layers.append(m.to(dev).eval())
    fc_in, fc_out = layers
    torch.manual_seed(0)
    xs = [torch.randn(8, M // 8, E, device=dev) for _ in range(2)]
    pkg.calibrate_cpt_layer(fc_in, bits, xs)
    with torch.no_grad():
        hs = [F.gelu(fc_in(x)) for x in xs]
    pkg.calibrate_cpt_layer(fc_out, bits, hs)
    del hs
    x = xs[0]
    flop = 2 * M * 2 * (E * H + E * r + r * H)
    with torch.no_grad():
        t_two = timeit(lambda: fc_out(F.gelu(fc_in(x))), 30)
        t_fused = timeit(lambda: pkg.cpt_mlp_forward(fc_in, fc_out, x), 30)
        for l in layers: l.cache_operands = False
        t_two_rq = timeit(lambda: fc_out(F.gelu(fc_in(x))), 30)
        t_fused_rq = timeit(lambda: pkg.cpt_mlp_forward(fc_in, fc_out, x), 30)
    print(f'CPT feed-forward {qt}{bits} {M} tokens E={E}: two layers + gelu {t_two*1e3:.4f} ms ({flop/t_two/1e12:.0f} TFLOP/s), fused pair '
          f'{t_fused*1e3:.4f} ms ({flop/t_fused/1e12:.0f} TFLOP/s); weights re-quantized every call: {t_two_rq*1e3:.4f} vs {t_fused_rq*1e3:.4f} ms', flush=True)
