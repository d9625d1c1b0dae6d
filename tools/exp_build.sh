#!/bin/bash
# usage: tools/exp_build.sh "-DFOO -DBAR"   -- rebuild maxpath.hip with experiment macros (development aid)
cd "$(dirname "$0")/../aligner_amd/csrc"
touch maxpath.hip
make -s all CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-function -I../../include -I. $1" 2>&1 | grep -v "warning\|^ *[0-9]* *|\|\^\|generated" | head
