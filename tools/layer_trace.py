"""Per-kernel time of one SPLinearWithLoRA forward shape under rocprofv3 (run: rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/layer_trace.py M K N r bits qtype [cached])."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from config_bench import build

M, K, N, r, bits = (int(v) for v in sys.argv[1:6])
qt = sys.argv[6]
layer, x = build(M, K, N, r, bits, qt, True)
layer.cache_operands = len(sys.argv) > 7 and sys.argv[7] == "cached"
with torch.no_grad():
    for _ in range(60):
        layer(x)
torch.cuda.synchronize()
print("path", layer._last_path)
