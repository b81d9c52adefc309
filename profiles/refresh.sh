#!/bin/bash
# Re-creates the judged artefacts of profiles/ on the GPU box (run from the repo root through gpurun); everything is
# written under gpurun_out/refresh/ and copied into profiles/ afterwards (see the end of this script's caller).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 bench.py --path f32 --steps 50 --no-cpu-baseline > $OUT/f32path_bench.json 2>> $OUT/bench.err; echo "bench f32 rc=$?"
python3 tools/config_bench.py --out $OUT/configs.json > $OUT/configs.log 2>&1; echo "configs rc=$?"
python3 tools/train_step_bench.py > $OUT/train_step.txt 2>&1; echo "train rc=$?"
python3 tools/cpt_bench.py > $OUT/cpt.txt 2>&1; echo "cpt rc=$?"
(python3 tools/block_bench.py; python3 tools/spblock_bench.py; python3 tools/spblock_bench.py --batch 8) > $OUT/blocks.txt 2>&1; echo "blocks rc=$?"
(python3 tools/gpt2_forward_bench.py; python3 tools/gpt2_forward_bench.py --layers 24 --embd 1024 --heads 16 --bits 6 --qtype log --batch 64) 2>/dev/null | grep "^{" > $OUT/gpt2_forward.jsonl; echo "gpt2 rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o f16x2 -- python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err; echo "trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -o f32 -- python3 $ROOT/bench.py --path f32 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_f32_bench.json 2>> $OUT/trace.err; echo "trace f32 rc=$?"
cd $ROOT
bash profiles/run_pmc.sh refresh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
python3 profiles/pmc_summary.py gpurun_out/pmc_refresh > $OUT/pmc_summary.txt 2>&1
echo done
