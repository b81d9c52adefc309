#!/bin/bash
# How long must the untimed setup phase of bench.py be before the FIRST timed region is in steady state?  (run on the GPU box)
for s in ${@:-200 2000 200 2000}; do
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --setup-steps $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('setup', $s, 'ms/step', d['ms_per_step'], 'repeats median/min', d['ms_per_step_stats']['median'], d['ms_per_step_stats']['min'], 'kernel', d['roofline']['kernel_ms_avg'])"
done
