for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['with_cached_weight_operands']['ms_per_step'])"
  SPQ_LIB=$PWD/tools/libvariants/libspq_r02.so python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02', d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['with_cached_weight_operands']['ms_per_step'])"
done
