#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]: builds the working tree's library into build/abl/<name>/ (A/B timing on one box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
n=$1; shift
mkdir -p $R/build/abl/$n
cd $R/grok_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable "$@" -shared -o $R/build/abl/$n/libgrok_amd.so \
  context.hip kernels_ingest.hip kernels_dwt.hip kernels_ht.hip kernels_htdec.hip kernels_t1dec.hip kernels_t1lanes.hip kernels_t2.hip kernels_idwt.hip ../../build/source_stamp.cpp node.cpp geometry.cpp t2_writer.cpp image.cpp
echo built $R/build/abl/$n/libgrok_amd.so
