"""BASELINE config 3: one full SPBlock (LayerNorm -> c_attn -> causal attention -> c_proj -> residual -> LayerNorm -> c_fc ->
GELU -> c_proj -> residual), 4-bit minmax per-channel + LoRA r = 64, batch x 1024 tokens, on one GPU.  The four linears, the
LayerNorms and the GELU run on this library's kernels; attention and the residual adds are stock torch-ROCm ops."""
import argparse, os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
dev = 'cuda:0'
ap = argparse.ArgumentParser(); ap.add_argument('--batch', type=int, default=32); ap.add_argument('--seq', type=int, default=1024)
ap.add_argument('--embd', type=int, default=768); ap.add_argument('--heads', type=int, default=12); ap.add_argument('--bits', type=int, default=4)
args = ap.parse_args()
E, bits, r = args.embd, args.bits, 64
cfg = types.SimpleNamespace(n_embd=E, n_head=args.heads, n_positions=args.seq, layer_norm_epsilon=1e-5, bit_widths=[bits, 32],
                            lora_rank_per_bit={bits: r, 32: 0}, lora_alpha_per_bit={bits: 64, 32: 0},
                            quantizer_per_bit={bits: 'minmax', 32: None}, per_channel_quantization=True)
torch.manual_seed(0)
blk = pkg.SPBlock(cfg, bit_widths=[bits, 32])
with torch.no_grad():
    for n, p in blk.named_parameters():
        if p.dim() > 1 and 'lora_B' not in n and 'lora_A' not in n: p.normal_(0, 0.02)
        elif 'lora_B' in n: p.normal_(0, 0.01)
blk = blk.to(dev).eval()
g = torch.Generator(device='cpu').manual_seed(1)
mk = lambda: torch.randn(args.batch, args.seq, E, generator=g).to(dev)
pkg.calibrate_model(blk, bits, [mk(), mk()])
x = mk()
M = args.batch * args.seq


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


lin_flop = M * (2 * E * 12 * E + 2 * r * (2 * E + 3 * E + 2 * E + 5 * E + 5 * E - 2 * E))   # 2MKN + 2Mr(K+N) over the four linears
lin_flop = sum(2 * M * (K * N + K * r + r * N) for K, N in ((E, 3 * E), (E, E), (E, 4 * E), (4 * E, E)))
lins = [blk.attn.c_attn, blk.attn.c_proj, blk.mlp.c_fc, blk.mlp.c_proj]
with torch.no_grad():
    for l in lins: l.fuse_norm = False
    t_blk_sep = timeit(lambda: blk(x))                     # LayerNorm as its own kernel in front of c_attn / c_fc
    t_pairs_sep = timeit(lambda: blk.attn.c_attn(blk.ln_1(x))) + timeit(lambda: blk.mlp.c_fc(blk.ln_2(x), activation='gelu'))
    for l in lins: l.fuse_norm = True
    t_pairs_fused = timeit(lambda: blk.attn.c_attn(x, pre_norm=blk.ln_1)) + timeit(lambda: blk.mlp.c_fc(x, activation='gelu', pre_norm=blk.ln_2))
    t_blk = timeit(lambda: blk(x))
    h1 = blk.ln_1(x)
    t_ln = timeit(lambda: blk.ln_1(x))
    t_lin = (timeit(lambda: blk.attn.c_attn(h1)) + timeit(lambda: blk.attn.c_proj(h1)) + timeit(lambda: blk.mlp(h1)))
    t_attn = timeit(lambda: blk.attn(h1))
print(f'SPBlock {args.batch} x {args.seq} tokens, E={E}, {bits}-bit minmax + LoRA r=64: block forward {t_blk:.3f} ms '
      f'({lin_flop / t_blk / 1e9:.0f} TFLOP/s counting the four linears only); of which: 4 linears + GELU {t_lin:.3f} ms '
      f'({lin_flop / t_lin / 1e9:.0f} TFLOP/s), 2 LayerNorms {2 * t_ln:.3f} ms, attention incl. its two linears {t_attn:.3f} ms; '
      f'LayerNorm inside the activation pass (default) vs as its own kernel: block {t_blk:.3f} vs {t_blk_sep:.3f} ms, '
      f'[ln_1 -> c_attn] + [ln_2 -> c_fc + GELU] {t_pairs_fused:.3f} vs {t_pairs_sep:.3f} ms')
