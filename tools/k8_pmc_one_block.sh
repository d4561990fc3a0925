#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set -d /tmp/q_$i -o p --output-format csv -- python $R/tools/k8_one_block.py > /tmp/q_$i.log 2>&1
  f=$(find /tmp/q_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "t1_dec_kernel" in r["Kernel_Name"]:
        print(r["Counter_Name"], r["Counter_Value"])
PY
  else tail -3 /tmp/q_$i.log; fi
done
