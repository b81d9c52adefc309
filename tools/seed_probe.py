import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import llm_qat_on_gpt2_amd as pkg
from oracle import ref_cpu as O
dev = 'cuda:0'
def worst(y, ref, rel=1e-5):
    y, ref = y.double().cpu(), ref.double()
    rms = float(ref.pow(2).mean().sqrt())
    return float(((y - ref).abs() / (rel * ref.abs() + rel * rms)).max())
for (M, K, N, r, bits) in [(512, 256, 256, 32, 12), (4096, 768, 3072, 64, 8)]:
    for seed in range(0, 1000, 83):
        W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=seed, batch=4)
        ol = O.build_calibrated_layer(W, bias, A, B, [x0, x1], bits, 'minmax', True, r, r)
        layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: r, 32: 0}, {bits: 'minmax', 32: None})
        key = f'{bits}bit'
        with torch.no_grad():
            layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
            layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
        layer = layer.to(dev).eval(); layer.set_precision(bits)
        pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
        ref = ol.forward(x1)
        with torch.no_grad():
            w1 = worst(layer(x1.to(dev)), ref)
            layer.operand_path = pkg._lib.PATH_F32
            w2 = worst(layer(x1.to(dev)), ref)
        print(f'M={M} K={K} bits={bits} seed={seed}: f16 path err/bound {w1:.3f}   f32 path {w2:.3f}', flush=True)
