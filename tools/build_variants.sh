#!/bin/bash
# Builds libspq.so variants with extra -D flags into tools/libvariants/ (kernel tuning only; SPQ_LIB=<path> selects one).
#   tools/build_variants.sh name1 "-DFOO=1" name2 "-DBAR=2" ...
set -e
cd "$(dirname "$0")/../llm-qat-on-gpt2_amd/csrc"
mkdir -p ../../tools/libvariants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c spq_f16x2.hip -o /tmp/spq_f16x2_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 spq_api.o spq_quant.o spq_gemm_f32.o /tmp/spq_f16x2_$name.o spq_comm.o spq_gemm_tn.o -ldl \
      -o ../../tools/libvariants/libspq_$name.so && echo built $name ) &
done
wait
