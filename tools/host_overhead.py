import sys, time, torch
sys.path.insert(0, '/root/repo')
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O          # seeded input generator
dev = 'cuda:0'
K, N, r, bits = 768, 3072, 64, 4
W, bias, A, B, x0, x1 = O.make_workload(64, K, N, r, seed=0, batch=1)
layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: 'minmax', 32: None})
with torch.no_grad():
    layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
    layer.lora_adapters['4bit'].lora_A.copy_(A); layer.lora_adapters['4bit'].lora_B.copy_(B)
layer = layer.to(dev).eval(); layer.set_precision(bits)
pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
x = x0.to(dev)
for cache in (False, True):
    layer.cache_operands = cache
    with torch.no_grad():
        for _ in range(20): layer(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500): layer(x)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print(f"cache_operands={cache}: host {t_host/500*1e6:.1f} us per forward (enqueue only), {t_all/500*1e6:.1f} us incl. drain")
