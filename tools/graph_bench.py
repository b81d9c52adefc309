"""Eager against HIP-graph replay (llm_qat_on_gpt2_amd.GraphedForward) of one SPBlock forward at small token counts, where
launches and the Python between them, not the kernels, set the pace.   python tools/graph_bench.py"""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
dev, E, bits = 'cuda:0', 768, 4
cfg = types.SimpleNamespace(n_embd=E, n_head=12, n_positions=1024, layer_norm_epsilon=1e-5, bit_widths=[bits, 32],
                            lora_rank_per_bit={bits: 64, 32: 0}, lora_alpha_per_bit={bits: 64, 32: 0},
                            quantizer_per_bit={bits: 'minmax', 32: None}, per_channel_quantization=True)
torch.manual_seed(0)


class Stack(torch.nn.Module):
    def __init__(self, n):
        super().__init__(); self.h = torch.nn.ModuleList([pkg.SPBlock(cfg, bit_widths=[bits, 32]) for _ in range(n)])
    def forward(self, x):
        for b in self.h: x = b(x)
        return x


def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


model = Stack(12)
with torch.no_grad():
    for n, p in model.named_parameters():
        if 'lora_B' in n: p.normal_(0, 0.01)
        elif p.dim() > 1 and 'lora_A' not in n: p.normal_(0, 0.02)
model = model.to(dev).eval()
for b in model.h: b.set_precision(bits)
pkg.calibrate_model(model, bits, [torch.randn(4, 1024, E, device=dev) for _ in range(2)])
for (B, T) in [(1, 128), (1, 1024), (4, 1024), (8, 1024)]:
    x = torch.randn(B, T, E, device=dev)
    with torch.no_grad():
        want = model(x).clone()
        t_eager = timeit(lambda: model(x))
        g = pkg.GraphedForward(model, x)
        same = torch.equal(g(x), want)
        t_graph = timeit(lambda: g(x))
    print(f'12 x SPBlock, {B} x {T} tokens, 4-bit: eager {t_eager*1e3:.3f} ms, graph replay {t_graph*1e3:.3f} ms ({t_eager/t_graph:.2f}x), '
          f'{B*T/t_graph/1e3:.0f} k tokens/s, same output {same}', flush=True)
