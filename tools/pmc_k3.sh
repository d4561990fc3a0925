#!/bin/bash
# SQ / LDS / TCP counter passes over the 8K encode with every kernel alone (run on the GPU box); prints per-kernel-family means
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_k3
export GRK_AMD_OVERLAP=0 PROF_DECODE=${PROF_DECODE:-0} PROF_N=4
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
P3="SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR"
P4="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  timeout 150 rocprofv3 --pmc $P -d /tmp/pk$i -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pk$i.log 2>&1
  f=$(find /tmp/pk$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_k3/pass$i.csv || tail -3 /tmp/pk$i.log
  i=$((i+1))
done
python3 - <<'PY'
import csv, collections, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_k3/pass*.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        inv = "idwt" in k
        top = "dwt_level_kernel<false, 3" in k or "dwt53_pk_kernel<3" in k
        s=("ht_fallback" if "ht_encode_fallback" in k else "ht" if "ht_encode" in k else "idwt0" if inv and top else "idwt" if inv
           else "dwt0" if top else "dwtN" if "dwt_level" in k or "dwt53_pk" in k else "prep" if "ht_dec_prep" in k else "vlc" if "ht_dec_vlc" in k else "ms" if "ht_dec_ms" in k else None)
        if s: acc[(s,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (s,c),v in sorted(acc.items()):
        print("%-12s %-26s mean %14.1f  launches %d  sum/4 steps %14.1f" % (s,c,sum(v)/len(v),len(v),sum(v)/4))
PY
