#!/bin/bash
# PMC passes over tools/pp_bench (separate runs per counter group, no tracing flags); summaries under gpurun_out/pp_pmc/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
BIN=${1:-tools/pp_bench}
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pp_pmc/$name -- $ROOT/$BIN > $ROOT/gpurun_out/pp_pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for name in ("sq1", "sq2", "tcc"):
    files = glob.glob(f"gpurun_out/pp_pmc/{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
