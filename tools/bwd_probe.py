import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O          # seeded input generator
dev = 'cuda:0'
M, K, N, r, bits, qt = 8192, 768, 3072, 64, 4, 'minmax'
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
for _ in range(20):
    x.grad = None; layer(x).backward(g)
torch.cuda.synchronize()
