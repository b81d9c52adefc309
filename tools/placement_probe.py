"""Does the forward's time depend on WHERE its buffers land?  One process, several trials; each trial first allocates a dummy block of another size (kept
alive), so that the caching allocator hands the layer's operands, workspace and output different addresses; 300 forwards timed per trial."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from config_bench import build
keep = []
for trial, pad_mb in enumerate((0, 3, 17, 64, 129, 0, 511, 5)):
    if pad_mb:
        keep.append(torch.empty(pad_mb * 1024 * 1024 + 4096 * trial, dtype=torch.uint8, device='cuda:0'))
    torch.cuda.empty_cache() if trial == 5 else None
    layer, x = build(8192, 768, 3072, 64, 4, 'minmax', True)
    layer.cache_operands = False
    with torch.no_grad():
        for _ in range(300): y = layer(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300): y = layer(x)
        b.record(); b.synchronize()
    print(f"trial {trial} (dummy {pad_mb} MB before it): {a.elapsed_time(b) / 300 * 1e3:.1f} us per forward; y at 0x{y.data_ptr():x}, x at 0x{x.data_ptr():x}", flush=True)
    del layer, x, y
