#!/bin/bash
# scratch: one GPU-box call of round 5 (edited per call)
cd $GRAFT_REPO_ROOT
echo "=== which K3 launches leave room (GRK_AMD_K3_ROOM: bit 0 top class, bit 1 the rest), and how much (r119: 120 registers = 4 waves, 32 free; r135: 136 = 3 waves)"
timeout 1500 python tools/k3_ab.py --rounds 2 head@GRK_AMD_K3_ROOM=0 head@GRK_AMD_K3_ROOM=1 head@GRK_AMD_K3_ROOM=2 head@GRK_AMD_K3_ROOM=3 r135@GRK_AMD_K3_ROOM=3 r119@GRK_AMD_K3_ROOM=3 2>&1 | grep -v amdgpu | tail -n 8
