# A/B of library variants on one box:  bash tools/ab_lib.sh "" tools/libvariants/libspq_x.so ...   ("" = the in-tree library)
for i in 1 2; do for v in "$@"; do SPQ_LIB=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err && python -c "
import json,sys
d=json.load(open('gpurun_out/ab.json')); print('${v:-in-tree}', d['ms_per_step'], d['ms_per_step_stats']['median'], d['roofline']['kernel_ms_avg'], d['with_cached_weight_operands']['ms_per_step'])
"; done; done
