"""SURVEY.md 8(d) configs 2-5 on one GPU: per-layer forward time of SPLinearWithLoRA, warm (back-to-back) and cold
(L2/MALL flushed with a 512 MB write before every launch), median and min over HIP-event timed iterations.
    python tools/config_bench.py [--out profiles/r01_configs.json]
Linear layers only (LN / attention / GELU are stock torch-ROCm and not on the path)."""
import argparse, json, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as O          # seeded input generator

dev = 'cuda:0'
PATHN = {1: 'f32', 2: 'f16x2', 4: 'f16x3', 5: 'i8'}


def build(M, K, N, r, bits, qt, pc, seed=0):
    W, bias, A, B, x0, x1 = O.make_workload(M, K, N, r, seed=seed, batch=max(1, M // 1024))
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: qt, 32: None}, per_channel=pc)
    key = f'{bits}bit'
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(dev).eval(); layer.set_precision(bits)
    pkg.calibrate_layer(layer, bits, [x0.to(dev), x1.to(dev)])
    return layer, x0.to(dev)


def timed(layer, x, iters, flush=None):
    ts = []
    with torch.no_grad():
        for _ in range(5): layer(x)
        for _ in range(iters):
            if flush is not None: flush.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); layer(x); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
    return ts


def warm_loop(layer, x, iters):
    with torch.no_grad():
        for _ in range(5): layer(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): layer(x)
        b.record(); b.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--out', default='gpurun_out/configs.json')
    ap.add_argument('--configs', default='2,3,4,5', help='comma-separated subset of SURVEY 8(d) configs'); args = ap.parse_args()
    flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)
    S, MED = 768, 1024
    cases = [  # config, name, M, K, N, r, bits, qtype, per_channel, repeats of this layer in the config
        ('2', 'c_fc 8-bit minmax per-channel', 4096, S, 4 * S, 64, 8, 'minmax', True, 1),
        ('2', 'c_fc 8-bit minmax per-tensor', 4096, S, 4 * S, 64, 8, 'minmax', False, 1),
        ('3', 'c_attn 4-bit', 32768, S, 3 * S, 64, 4, 'minmax', True, 1),
        ('3', 'attn c_proj 4-bit', 32768, S, S, 64, 4, 'minmax', True, 1),
        ('3', 'c_fc 4-bit', 32768, S, 4 * S, 64, 4, 'minmax', True, 1),
        ('3', 'mlp c_proj 4-bit', 32768, 4 * S, S, 64, 4, 'minmax', True, 1),
        ('4', 'c_attn 4-bit (x12 layers x32 micro-batches of 8x1024)', 8192, S, 3 * S, 64, 4, 'minmax', True, 12 * 32),
        ('4', 'attn c_proj 4-bit', 8192, S, S, 64, 4, 'minmax', True, 12 * 32),
        ('4', 'c_fc 4-bit', 8192, S, 4 * S, 64, 4, 'minmax', True, 12 * 32),
        ('4', 'mlp c_proj 4-bit', 8192, 4 * S, S, 64, 4, 'minmax', True, 12 * 32),
        ('5', 'medium c_attn log 6-bit (x24 layers)', 8192, MED, 3 * MED, 64, 6, 'log', True, 24),
        ('5', 'medium attn c_proj log 6-bit', 8192, MED, MED, 64, 6, 'log', True, 24),
        ('5', 'medium c_fc log 6-bit', 8192, MED, 4 * MED, 64, 6, 'log', True, 24),
        ('5', 'medium mlp c_proj log 6-bit', 8192, 4 * MED, MED, 64, 6, 'log', True, 24),
    ]
    rows = []
    cases = [c for c in cases if c[0] in args.configs.split(',')]
    for cfg, name, M, K, N, r, bits, qt, pc, rep in cases:
        layer, x = build(M, K, N, r, bits, qt, pc)
        flop = 2 * M * (K * N + K * r + r * N)
        byts = 4 * (M * K + N * K + K * r + r * N + N + M * N)
        res = {'config': cfg, 'layer': name, 'M': M, 'K': K, 'N': N, 'r': r, 'bits': bits, 'qtype': qt, 'per_channel': pc,
               'flop': flop, 'algorithmic_bytes': byts, 'repeats_in_config': rep}
        for mode, cache in (('requantize_every_call', False), ('cached_weight_operands', True)):
            layer.cache_operands = cache
            w = warm_loop(layer, x, 50)
            cold = timed(layer, x, 20, flush)
            res[mode] = {'warm_ms': round(w, 4), 'warm_TFLOPs': round(flop / w / 1e9, 1),
                         'cold_ms_median': round(statistics.median(cold), 4), 'cold_ms_min': round(min(cold), 4),
                         'cold_TFLOPs_median': round(flop / statistics.median(cold) / 1e9, 1)}
        res['operand_path'] = PATHN[layer._last_path]
        rows.append(res)
        print(json.dumps(res), flush=True)
        del layer, x
        torch.cuda.empty_cache()
    tot = {}
    for r_ in rows:
        t = tot.setdefault(r_['config'], {'flop': 0, 'ms_warm_requant': 0.0, 'ms_warm_cached': 0.0})
        t['flop'] += r_['flop'] * r_['repeats_in_config']
        t['ms_warm_requant'] += r_['requantize_every_call']['warm_ms'] * r_['repeats_in_config']
        t['ms_warm_cached'] += r_['cached_weight_operands']['warm_ms'] * r_['repeats_in_config']
    for c, t in tot.items():
        t['TFLOPs_requant'] = round(t['flop'] / t['ms_warm_requant'] / 1e9, 1)
        t['TFLOPs_cached'] = round(t['flop'] / t['ms_warm_cached'] / 1e9, 1)
    out = {'device': torch.cuda.get_device_name(0), 'rows': rows, 'linear_only_totals_one_gpu': tot,
           'note': 'config 4/5 totals = per-layer warm time x (layers x micro-batches) on ONE GPU; config 2 rows are separate variants'}
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
    print(json.dumps(tot))


if __name__ == '__main__':
    main()
