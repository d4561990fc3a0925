#!/bin/bash
# SQ counter passes over the 8K encode (run on the GPU box; writes gpurun_out/pmc_sq_*.csv)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_FLAT SQ_INSTS_VMEM_WR"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 200 rocprofv3 --pmc $P -d /tmp/pmc$i -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_sq_$i.csv || tail -5 /tmp/pmc$i.log
  i=$((i+1))
done
python - <<'PY'
import csv, collections, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_sq_*.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        s=("ht" if "ht_encode" in k else "dwt0" if "dwt_level_kernel<false, 3, 1>" in k else "dwtN" if "dwt_level" in k and "idwt" not in k
           else "idwt" if "idwt_level" in k else "vlc" if "ht_dec_vlc" in k else "ms" if "ht_dec_ms" in k else None)
        if s: acc[(s,r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (s,c),v in sorted(acc.items()):
        print(s,c,sum(v)/len(v),len(v))
PY
