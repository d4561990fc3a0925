#!/bin/bash
# scratch: one GPU-box call of round 5 (edited per call)
cd $GRAFT_REPO_ROOT
A=$PWD/build/abl
echo "=== k3_ab (md5 / lengths of the stop builds are wrong by design)"; timeout 900 python tools/k3_ab.py --rounds 1 head stop4 stop1 stop2 stop3 2>&1 | tail -n 7
for v in head stop4 stop1 stop2 stop3; do lib=$A/$v/libgrok_amd.so; [ $v = head ] && lib=$PWD/grok_amd/lib/libgrok_amd.so; echo "=== per-block counters $v"; GRK_AMD_LIB=$lib bash tools/pmc_k3_short.sh 2>&1 | tail -n 2; done
