"""Kernel timeline from a rocprofv3 --kernel-trace csv (dev tool): python tools/timeline.py kernel_trace.csv [last N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("grk_amd::(anonymous namespace)::", "")[:46]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-46s q%-3s start %9.3f us  end %9.3f us  dur %8.3f us" % (name, r.get("Queue_Id", "?"), s / 1e3, e / 1e3, (e - s) / 1e3))
