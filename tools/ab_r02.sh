#!/bin/bash
# A/B of the current library against the round-2 library on ONE box (boxes of the pool differ by ~8 %).  Build the old one first, in the
# build container:   rm -rf /tmp/r02src && mkdir -p /tmp/r02src && git archive 45c5a10 llm-qat-on-gpt2_amd/csrc include | tar -x -C /tmp/r02src &&
#                    make -C /tmp/r02src/llm-qat-on-gpt2_amd/csrc && cp /tmp/r02src/llm-qat-on-gpt2_amd/libspq.so tools/libvariants/libspq_r02.so
# then, on the GPU box:   bash tools/ab_r02.sh
exec bash tools/ab_libs.sh llm-qat-on-gpt2_amd/libspq.so tools/libvariants/libspq_r02.so
