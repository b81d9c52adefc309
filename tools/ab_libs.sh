#!/bin/bash
# A/B of library builds on ONE box, interleaved: bash tools/ab_libs.sh <lib.so> <lib.so> ...   (paths relative to the repo root)
for i in 1 2 3; do
  for lib in "$@"; do
    SPQ_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'], 'cached', d['with_cached_weight_operands']['ms_per_step'])"
  done
done
