#!/bin/bash
# run tools/stamps.py against every experiment build under aligner_amd/lib/exp (development aid)
for f in aligner_amd/lib/exp/*/libaligner_amd.so; do
  echo "== $f"
  ALIGNER_AMD_LIB=$PWD/$f timeout -k 10 120 python tools/stamps.py "$@" 2>&1 | grep -v amdgpu.ids | grep "fwd_done:\|own stream\|walk_done:\|end:\|first workgroup"
done
