import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O
dev = 'cuda:0'
M, K, N, r, bits = 8192, 768, 768, 64, 4
W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=0, batch=8)
layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: 'minmax', 32: None})
with torch.no_grad():
    layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
    layer.lora_adapters['4bit'].lora_A.copy_(A); layer.lora_adapters['4bit'].lora_B.copy_(B)
layer = layer.to(dev).eval(); layer.set_precision(bits)
pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
x = x0.to(dev)
layer.cache_operands = False
with torch.no_grad():
    for _ in range(20): layer(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): layer(x)
    pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(18)
