import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O
dev = 'cuda:0'
for (M, K, N, r, bits) in [(4096, 768, 3072, 64, 8), (512, 256, 256, 32, 12), (8192, 3072, 768, 64, 4)]:
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=3, batch=4)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: 'minmax', 32: None})
    key = f'{bits}bit'
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(dev).eval(); layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
    x = x1.to(dev)
    with torch.no_grad():
        y0 = layer(x).clone()
        bad = 0
        for i in range(300):
            layer.cache_operands = (i % 2 == 0)
            y = layer(x)
            if not torch.equal(y, y0):
                bad += 1
                if bad <= 3:
                    d = (y - y0).abs()
                    print('  diff at iter', i, 'max', d.max().item(), 'count', int((d > 0).sum()), 'rows', torch.nonzero(d.amax(dim=-1).reshape(-1) > 0).flatten()[:8].tolist())
    print(f'M={M} K={K} N={N} bits={bits} lora_down_f16={os.environ.get("SPQ_LORA_DOWN_F16","0")}: {bad} of 300 runs differ', flush=True)
