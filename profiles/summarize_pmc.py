#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots prescribes).

Units and corrections (same guide):
  * FETCH_SIZE / WRITE_SIZE are in KiB  -> x 1024
  * on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read -> x 2.
    Calibration on our own access patterns: ingest_kernel reads 201,326,592 B of u8 pixels (4 B/lane)
    and reports 98,332 KiB = 100.7 MB (ratio 0.5001); ht_encode_kernel reads 805.3 MB of int32
    (8 B/lane) and reports 398,769 KiB (x2 = 816.7 MB, +1.4 % over the algorithmic bytes).
  * WRITE_SIZE needs no correction: ingest_kernel writes 805,306,368 B and reports 786,432.0 KiB exactly.

usage: summarize_pmc.py fetch_counter_collection.csv write_counter_collection.csv out.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    for k in ("ht_encode_kernel", "idwt_level_kernel", "dwt_level_kernel", "ingest_kernel", "ht_dec_vlc_kernel",
              "ht_dec_ms_kernel", "egress_kernel"):
        if k in name:
            return k
    return None


def main(fetch_csv, write_csv, out_json):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for name in f:
        k = short(name)
        if not k or name not in w:
            continue
        fr = sum(f[name]) / len(f[name]) * 1024 * 2       # KiB -> B, gfx950 half-count correction
        wr = sum(w[name]) / len(w[name]) * 1024
        out[k] = {"launches_sampled": len(f[name]), "hbm_read_bytes_per_launch": int(fr),
                  "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(fr + wr),
                  "fetch_size_kib_raw_mean": sum(f[name]) / len(f[name]),
                  "write_size_kib_raw_mean": sum(w[name]) / len(w[name])}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main(*sys.argv[1:4])
