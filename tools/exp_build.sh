#!/bin/bash
# usage: tools/exp_build.sh TAG "-DFOO -DBAR"   -- build an EXPERIMENT copy of the library with extra macros
# (development aid).  The result goes to aligner_amd/lib/exp/TAG/libaligner_amd.so; the shipped
# aligner_amd/lib/libaligner_amd.so is never touched.  Select an experiment build with
#   ALIGNER_AMD_LIB=$PWD/aligner_amd/lib/exp/TAG/libaligner_amd.so python ...      (tools/exp_run.sh does)
set -e
TAG=${1:?tag}; shift
cd "$(dirname "$0")/../aligner_amd/csrc"
make -s all OUTDIR=../lib/exp/$TAG CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-function -I../../include -I. $*" 2>&1 | grep -v "warning\|^ *[0-9]* *|\|\^\|generated" | head
