#!/bin/bash
# FETCH_SIZE calibrated on the access widths the DWT kernels use: tools/mem_width_bench streams 512 MB per launch
# with 2-, 4-, 8-, 16-byte accesses per lane; one rocprofv3 --pmc pass per counter.  Output: gpurun_out/fetch_cal/*.csv + a summary.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/mem_width_bench tools/mem_width_bench.hip
export TMPDIR=/tmp
mkdir -p gpurun_out/fetch_cal
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d gpurun_out/fetch_cal/$c -o p --output-format csv -- ./tools/mem_width_bench > gpurun_out/fetch_cal/$c.log 2>&1 || true
done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    hits = glob.glob("gpurun_out/fetch_cal/%s/**/*counter_collection.csv" % c, recursive=True)
    if not hits: print(c, "no csv"); continue
    tot = collections.defaultdict(list)
    for r in csv.DictReader(open(hits[0])):
        if r["Counter_Name"] == c: tot[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in tot.items():
        print("%-10s %-60s launches %3d  mean %.1f KiB = %.1f MB" % (c, k[:60], len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / 1e6))
PY
