#!/bin/bash
# PMC passes over `bench.py` (separate runs per counter group, no tracing flags), CSVs under gpurun_out/pmc_<tag>/
# usage (on the GPU box, from the repo root): bash profiles/run_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}/$name -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $ROOT/gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "pass $name rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
