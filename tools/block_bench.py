"""SURVEY.md §8 f1 at GPT-2-small dims (8 x 1024 tokens, E = 768, 4-bit minmax): the LayerNorm producer as one HIP pass vs the
reference's composed formula on stock torch-ROCm ops, and SPMLP with the GELU inside c_fc's store vs a separate F.gelu."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O
dev = 'cuda:0'


def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


E, M, bits, r = 768, 8192, 4, 64
x = torch.randn(8, M // 8, E, device=dev)
ln = pkg.SwitchableLayerNorm(E, precision_levels=[bits, 32]).to(dev)
ln.set_precision(bits)
w, b = ln.weights[str(bits)], ln.biases[str(bits)]
with torch.no_grad():
    t_k = timeit(lambda: ln(x))
    t_c = timeit(lambda: ln._composed(x, w, b))
    t_f = timeit(lambda: torch.nn.functional.layer_norm(x, (E,), w, b, 1e-5))
print(f'SwitchableLayerNorm [{M} x {E}]: HIP kernel {t_k:.4f} ms ({2 * 4 * M * E / t_k / 1e6:.0f} GB/s algorithmic), reference formula on '
      f'torch-ROCm ops {t_c:.4f} ms, F.layer_norm {t_f:.4f} ms')

cfg = types.SimpleNamespace(n_embd=E, bit_widths=[bits, 32], lora_rank_per_bit={bits: r, 32: 0}, lora_alpha_per_bit={bits: 64, 32: 0},
                            quantizer_per_bit={bits: 'minmax', 32: None}, per_channel_quantization=True)
m = pkg.SPMLP(cfg, bit_widths=[bits, 32])
key = f'{bits}bit'
Wf, bf, Af, Bf, x0, x1 = O.make_workload(M, E, 4 * E, r, seed=0, batch=8)
Wp, bp, Ap, Bp, _, _ = O.make_workload(M, 4 * E, E, r, seed=1, batch=8)
with torch.no_grad():
    for lin, (W, bb, A, B) in ((m.c_fc, (Wf, bf, Af, Bf)), (m.c_proj, (Wp, bp, Ap, Bp))):
        lin.linear.weight.copy_(W); lin.linear.bias.copy_(bb)
        lin.lora_adapters[key].lora_A.copy_(A); lin.lora_adapters[key].lora_B.copy_(B)
m = m.to(dev).eval()
pkg.calibrate_model(m, bits, [x0.to(dev), x1.to(dev)])
xin = x0.to(dev)
flop = 2 * M * (2 * E * 4 * E + 2 * (E + 4 * E) * r)
for cache in (True, False):
    m.c_fc.cache_operands = m.c_proj.cache_operands = cache
    with torch.no_grad():
        t_fused = timeit(lambda: m(xin))
        t_sep = timeit(lambda: m.c_proj(torch.nn.functional.gelu(m.c_fc(xin))))
        t_gelu = timeit(lambda: torch.nn.functional.gelu(m.c_fc._last_y)) if hasattr(m.c_fc, '_last_y') else None
    print(f'SPMLP {M} tokens, 4-bit minmax, {"cached operands" if cache else "weights re-quantized every call"}: fused GELU {t_fused:.4f} ms '
          f'({flop / t_fused / 1e9:.0f} TFLOP/s), separate F.gelu {t_sep:.4f} ms ({flop / t_sep / 1e9:.0f} TFLOP/s)')
