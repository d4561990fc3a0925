#!/bin/bash
# builds ablation variants of libgrok_amd.so (timing experiments only; results are wrong by design)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -shared"
S="grok_amd/csrc/context.hip grok_amd/csrc/kernels_ingest.hip grok_amd/csrc/kernels_dwt.hip grok_amd/csrc/kernels_ht.hip grok_amd/csrc/kernels_htdec.hip grok_amd/csrc/kernels_t1dec.hip grok_amd/csrc/kernels_idwt.hip grok_amd/csrc/geometry.cpp grok_amd/csrc/t2_writer.cpp"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}; [ "$defs" = "$name" ] && defs=""
  mkdir -p build/abl/$name
  /opt/rocm/bin/hipcc $F $defs -o build/abl/$name/libgrok_amd.so $S &
done
wait
