"""Capture one SPLinearWithLoRA forward (weights re-quantized inside) in a HIP graph and compare replay time with eager."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O
dev = 'cuda:0'
for (M, K, N) in [(8192, 768, 3072), (8192, 768, 768), (1024, 768, 3072)]:
    r, bits = 64, 4
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=0, batch=8)
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
            for _ in range(5): y_ref = layer(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): layer(x)
            torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 100
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): layer(x)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = layer(x)
            g.replay(); torch.cuda.synchronize()
            same = torch.equal(y, y_ref)
            t0 = time.perf_counter()
            for _ in range(100): g.replay()
            torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 100
        print(f'M={M} K={K} N={N} cached={cache}: eager {eager*1e3:.4f} ms, graph replay {graph*1e3:.4f} ms, same output {same}', flush=True)
