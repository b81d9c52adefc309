"""Forward + straight-through backward of one SPLinearWithLoRA at the headline shape (training mode: operands rebuilt
every step, LoRA factors and the input receive gradients, the base weight is frozen as in main_sp.py:83)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O          # seeded input generator
dev = 'cuda:0'
for (M, K, N, r, bits, qt) in [(8192, 768, 3072, 64, 4, 'minmax'), (8192, 3072, 768, 64, 4, 'minmax'), (8192, 1024, 4096, 64, 6, 'log')]:
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=0, batch=8)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: qt, 32: None})
    key = f'{bits}bit'
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(dev).train(); layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
    layer.linear.weight.requires_grad_(False); layer.linear.bias.requires_grad_(False)
    x = x0.to(dev).requires_grad_(True)
    g = torch.randn(8, M // 8, N, device=dev) * 1e-3
    flop_f = 2 * M * (K * N + K * r + r * N)
    res = {}
    for limbs in (True, False):
        layer.backward_limbs = limbs
        for _ in range(5):
            x.grad = None; layer(x).backward(g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            y = layer(x)
        torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 30
        t0 = time.perf_counter()
        for _ in range(30):
            x.grad = None; layer(x).backward(g)
        torch.cuda.synchronize(); tfb = (time.perf_counter() - t0) / 30
        res[limbs] = x.grad.clone()
        print(f'{qt}{bits} M={M} K={K} N={N} backward_limbs={limbs}: fwd {tf*1e3:.3f} ms, fwd+bwd {tfb*1e3:.3f} ms '
              f'(bwd {1e3*(tfb-tf):.3f} ms; fwd+bwd {3*flop_f/tfb/1e12:.1f} TFLOP/s counting bwd = 2 x fwd FLOPs... see DESIGN)', flush=True)
    d = (res[True] - res[False]).abs().max().item(); rms = res[False].pow(2).mean().sqrt().item()
    print(f'   max|grad_x(limbs) - grad_x(f32)| = {d:.3e}, rms {rms:.3e}, ratio {d/rms:.2e}')
