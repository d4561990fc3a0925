#!/bin/bash
# SQ counters of the PIPELINED 8K encode, on the GPU box: rocprofv3 --pmc passes with --kernel-trace so that the
# dispatch intervals say whether the kernels still overlapped under counter collection (dispatch counters are per kernel: the tool
# may serialise what the timed region overlaps -- the trace answers that), then per kernel family the counter sums per frame.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ovl; mkdir -p $OUT
export PROF_DECODE=0 PROF_N=12 PROF_PIPELINE=1
P1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_CYCLES"
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/po$i
  timeout 200 rocprofv3 --pmc $P --kernel-trace -d /tmp/po$i -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/po$i.log 2>&1 || tail -3 /tmp/po$i.log
  f=$(find /tmp/po$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/pass$i.csv
  f=$(find /tmp/po$i -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $OUT/trace$i.csv
  i=$((i+1))
done
# the same region without counters: the reference timeline
rm -rf /tmp/po0
timeout 200 rocprofv3 --kernel-trace -d /tmp/po0 -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/po0.log 2>&1
f=$(find /tmp/po0 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $OUT/trace0.csv
python3 - <<'PY'
import csv, collections, glob, os
R=os.environ["GRAFT_REPO_ROOT"]; OUT=R+"/gpurun_out/pmc_ovl"
def fam(k):
    return ("ht" if "ht_encode_kernel" in k else "ht_fallback" if "ht_encode_fallback" in k else
            "dwt0" if ("dwt53_pk_kernel<3" in k or "dwt_level_kernel<false, 3" in k) and "idwt" not in k else
            "dwtN" if ("dwt53_pk" in k or "dwt_level" in k) and "idwt" not in k else None)
for t in sorted(glob.glob(OUT+"/trace*.csv")):
    rows=[r for r in csv.DictReader(open(t)) if fam(r["Kernel_Name"])]
    iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),fam(r["Kernel_Name"])) for r in rows)
    if not iv: continue
    span=iv[-1][1]-iv[0][0]; busy=sum(e-s for s,e,_ in iv)
    ovl=0; last_end=iv[0][0]
    for s,e,_ in iv:
        if s<last_end: ovl+=min(e,last_end)-s
        last_end=max(last_end,e)
    per=collections.defaultdict(list)
    for s,e,f in iv: per[f].append((e-s)/1e3)
    print("%s: %d kernels, span %.3f ms, sum of durations %.3f ms, overlapped time %.3f ms" % (os.path.basename(t),len(iv),span/1e6,busy/1e6,ovl/1e6))
    for f,v in sorted(per.items()): print("    %-12s n %4d mean %9.1f us" % (f,len(v),sum(v)/len(v)))
for f in sorted(glob.glob(OUT+"/pass*.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        s=fam(r["Kernel_Name"])
        if s: acc[(s,r["Counter_Name"])].append(float(r["Counter_Value"]))
    n=int(os.environ.get("PROF_N","12"))
    for (s,c),v in sorted(acc.items()):
        print("%-12s %-24s per frame %16.1f  launches %d" % (s,c,sum(v)/n,len(v)))
PY
