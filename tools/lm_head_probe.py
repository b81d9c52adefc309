"""part2's lm_head is a CPTLinear n_embd -> vocab (50257, not a multiple of 4): f16-limb path (scalar-store epilogue) vs fp32 path."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O
dev = 'cuda:0'
M, K, N, r, bits = 4096, 768, 50257, 16, 4
W, bias, A, B, x0, x1 = O.make_cpt_workload(M, K, N, r, seed=0, batch=4)
m = pkg.CPTLinear(K, N, bit_widths=[bits, 32], quantizer_per_bit={bits: 'minmax', 32: None}, bias=False, shared_lora_rank=r, shared_lora_alpha=32)
with torch.no_grad():
    m.linear.weight.copy_(W); m.shared_lora.lora_A.copy_(A); m.shared_lora.lora_B.copy_(B)
m = m.to(dev).eval()
pkg.calibrate_cpt_layer(m, bits, [x0.to(dev), x1.to(dev)])
m.set_precision(bits)
x = x0.to(dev)
flop = 2 * M * (K * N + K * r + r * N)
with torch.no_grad():
    for _ in range(3): y = m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'CPT lm_head {K}->{N}, {M} tokens, 4-bit: {dt*1e3:.3f} ms ({flop/dt/1e12:.0f} TFLOP/s), path {m._last_path}')
