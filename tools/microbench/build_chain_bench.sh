#!/bin/bash
# builds tools/microbench/chain_bench.exe (timing build: s_memtime stamps of tile 37) from the repo root
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function ${CHAIN_DEFS:--DOJF_CHAIN_TIMING} -c tools/microbench/chain_bench.hip -o /tmp/cb.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/cb.o online_joint_depthfusion_and_semantic_amd/csrc/ojf_api.o -o tools/microbench/chain_bench.exe
