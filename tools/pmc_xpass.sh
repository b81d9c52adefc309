#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_xp/$name -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/pmc_xp_$name.log 2>&1
  echo "pass $name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_VMEM_WR
cd $ROOT && python3 profiles/pmc_summary.py gpurun_out/pmc_xp
