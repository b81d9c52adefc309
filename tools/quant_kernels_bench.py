"""HBM-bound kernels of rows a1/a3/a6 (statistics, fake-quant) and the front kernels of the fused forward, timed with HIP
events and priced against their algorithmic bytes (statistics 4 B/element, fake-quant 8 B/element)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
dev = 'cuda:0'
PEAK = 8000.0


def t_ms(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


rows = []


def report(name, nbytes, ms):
    gbs = nbytes / ms / 1e6
    rows.append({'kernel': name, 'algorithmic_MB': round(nbytes / 1e6, 2), 'us': round(ms * 1e3, 2), 'GBps': round(gbs, 0), 'frac_of_8TBps': round(gbs / PEAK, 3)})
    print(f'{name:58s} {nbytes / 1e6:8.2f} MB  {ms * 1e3:8.2f} us  {gbs:7.0f} GB/s  {gbs / PEAK:.3f} of HBM peak', flush=True)


x = torch.randn(8, 1024, 768, device=dev)
W = torch.randn(3072, 768, device=dev) * 0.02
with torch.no_grad():
    for qt in ('minmax', 'log'):
        for pc, tag in ((True, 'per-channel'), (False, 'per-tensor')):
            q = pkg.LearnableFakeQuantize(4 if qt == 'minmax' else 6, channel_dim=-1, quantizer_type=qt, per_channel=pc).to(dev)
            q.start_calibration()

            def stat():
                q.num_batches_collected = 0
                q(x)
            report(f'statistics x[8192,768] {qt} {tag} (incl. host shell)', 4 * x.numel(), t_ms(stat))
            q.finish_calibration()
            report(f'fake-quant x[8192,768] {qt} {tag}', 8 * x.numel(), t_ms(lambda: q(x)))
        qw = pkg.LearnableFakeQuantize(8, channel_dim=0, quantizer_type=qt, per_channel=False).to(dev)
        qw.start_calibration(); qw(W); qw.finish_calibration()
        report(f'fake-quant W[3072,768] {qt} 8-bit per-tensor (BASELINE config 1 on the GPU)', 8 * W.numel(), t_ms(lambda: qw(W)))
    ln = pkg.SwitchableLayerNorm(768, precision_levels=[4, 32]).to(dev)
    report('SwitchableLayerNorm x[8192,768]', 8 * x.numel(), t_ms(lambda: ln(x)))
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'quant_kernels.json'), 'w'), indent=1)
