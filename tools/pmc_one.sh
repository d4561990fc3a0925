#!/bin/bash
# one rocprofv3 --pmc pass over the 8K encode, kernels alone: tools/pmc_one.sh "COUNTER ..." -> per-family means
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GRK_AMD_OVERLAP=0 PROF_DECODE=${PROF_DECODE:-0} PROF_N=4
rm -rf /tmp/p1
timeout 150 rocprofv3 --pmc $1 -d /tmp/p1 -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/p1.log 2>&1
f=$(find /tmp/p1 -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { tail -3 /tmp/p1.log; exit 1; }
python3 - $f <<'PY'
import csv, collections, sys
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    s=("ht_fallback" if "ht_encode_fallback" in k else "ht" if "ht_encode" in k else "dwt0" if "dwt_level_kernel<false, 3" in k else "dwtN" if "dwt_level" in k and "idwt" not in k
       else "idwt" if "idwt_level" in k else "vlc" if "ht_dec_vlc" in k else "ms" if "ht_dec_ms" in k else None)
    if s: acc[(s,r["Counter_Name"])].append(float(r["Counter_Value"]))
for (s,c),v in sorted(acc.items()):
    print("%-12s %-26s per step %14.1f  (launches %d)" % (s,c,sum(v)/4,len(v)))
PY
