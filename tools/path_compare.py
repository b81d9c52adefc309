"""Wall time per forward of each operand path on a few workloads (GPU): python tools/path_compare.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O          # seeded input generator
dev = 'cuda:0'
L = pkg._lib
CASES = [  # M, K, N, r, bits, qtype, paths
    (8192, 768, 3072, 64, 4, 'minmax', ['f16x2', 'f16x3', 'f32']),
    (8192, 1024, 4096, 64, 6, 'log', ['f16x3', 'f32']),
    (8192, 4096, 1024, 64, 6, 'log', ['f16x3', 'f32']),
    (8192, 768, 3072, 64, 16, 'minmax', ['f16x3', 'f32']),
]
P = {'f16x2': L.PATH_F16X2, 'f16x3': L.PATH_F16X3, 'f32': L.PATH_F32}
for M, K, N, r, bits, qt, paths in CASES:
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=0, batch=8)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: qt, 32: None})
    key = f'{bits}bit'
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(dev).eval(); layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
    x = x0.to(dev)
    flop = 2 * M * (K * N + K * r + r * N)
    ys = {}
    for p in paths:
        layer.operand_path = P[p]
        with torch.no_grad():
            for _ in range(10): y = layer(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): y = layer(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        ys[p] = y
        print(f'{qt}{bits} M={M} K={K} N={N} {p:6s} {dt*1e3:.4f} ms  {flop/dt/1e12:.1f} TFLOP/s  path_used={layer._last_path}', flush=True)
    ref = ys['f32'].double()
    for p in paths[:-1]:
        d = (ys[p].double() - ref).abs().max().item()
        print(f'   max|{p} - f32| = {d:.3e}  (rms {ref.pow(2).mean().sqrt().item():.3e})')
