import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O
dev = 'cuda:0'
for (M, K, N) in [(8192, 768, 3072), (8192, 3072, 768), (32768, 3072, 768)]:
    r, bits = 64, 4
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=0, batch=8)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: 'minmax', 32: None})
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters['4bit'].lora_A.copy_(A); layer.lora_adapters['4bit'].lora_B.copy_(B)
    layer = layer.to(dev).eval(); layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
    x = x0.to(dev)
    with torch.no_grad():
        for _ in range(10): layer(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): layer(x)
        torch.cuda.synchronize()
    print(f'M={M} K={K} N={N} lora_down_f16={os.environ.get("SPQ_LORA_DOWN_F16","0")}: {(time.perf_counter()-t0)/50*1e3:.4f} ms', flush=True)
