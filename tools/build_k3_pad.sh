#!/bin/bash
# usage: tools/build_k3_pad.sh <name> <pad words> [extra flags]: the working tree's kernels_ht.hip with GRK_K3_PAD_WORDS, linked with the in-tree objects
# of everything else (build/obj, as __graft_entry__.build() left them) into build/abl/<name>/ -- the code-placement scan of r05
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
n=$1; w=$2; shift; shift
mkdir -p $R/build/abl/$n
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -x hip -c -o $R/build/abl/$n/kernels_ht.o $R/grok_amd/csrc/kernels_ht.hip
objs=$(ls $R/build/obj/*.o | grep -v kernels_ht.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/abl/$n/libgrok_amd.so $objs $R/build/abl/$n/kernels_ht.o
rm -f $R/build/abl/$n/kernels_ht.o
echo built $n
