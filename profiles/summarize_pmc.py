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

Kernel families are reported PER STEP (one encode_tiles / decode_tiles call): a family may be several
launches (DWT levels, the classes of the HT encoder; the last inverse DWT level carries K7 when fused); steps are counted by the one-per-step
kernels ht_alloc_init_kernel (encode) and ht_dec_vlc_kernel / t1_dec_kernel (decode).

usage: summarize_pmc.py fetch_counter_collection.csv write_counter_collection.csv out.json
"""
import collections
import csv
import json
import sys

FAMILIES = [  # (family, test on the kernel name); the inverse kernels first: "idwt_level_kernel" contains "dwt_level_kernel"
    # (r02: the packed 5/3 kernels dwt53_pk_kernel<NC, PX> / idwt53_pk_kernel<NC, PXO> belong to the same families)
    ("idwt_last_level_fused", lambda n: ("idwt_level_kernel<" in n and not n.split("idwt_level_kernel<")[1].startswith(("false, 1, 0", "true, 1, 0")))
                                        or ("idwt53_pk_kernel<" in n and not n.split("idwt53_pk_kernel<")[1].startswith("1, 0"))),
    ("idwt_level_kernel", lambda n: "idwt_level_kernel<" in n or "idwt53_pk_kernel<" in n),
    ("dwt_levels_1plus", lambda n: "dwt_level_kernel<false, 1, 0" in n or "dwt_level_kernel<true, 1, 0" in n or "dwt53_pk_kernel<1, 0" in n),
    ("dwt_level0_fused", lambda n: "dwt_level_kernel<" in n or "dwt53_pk_kernel<" in n),
    ("ht_encode_kernel", lambda n: "ht_encode_kernel" in n),
    ("ht_dec_prep_kernel", lambda n: "ht_dec_prep_kernel" in n),
    ("ht_dec_vlc_kernel", lambda n: "ht_dec_vlc_kernel" in n),
    ("ht_dec_ms_kernel", lambda n: "ht_dec_ms_kernel" in n),
    ("t1_dec_kernel", lambda n: "t1_dec_kernel" in n),
    ("t1_lanes_kernel", lambda n: "t1_lanes_kernel" in n),
    ("t1_recon_kernel", lambda n: "t1_recon_kernel" in n),
    ("egress_kernel", lambda n: "egress_kernel" in n),
    ("ingest_kernel", lambda n: "ingest_kernel" in n),
]


def family(name):
    for fam, test in FAMILIES:
        if test(name):
            return fam
    return None


def totals(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        cnt[name] += 1
        f = family(name)
        if f:
            tot[f] += float(r["Counter_Value"])
    return tot, cnt


def main(fetch_csv, write_csv, out_json):
    f, fc = totals(fetch_csv, "FETCH_SIZE")
    w, wc = totals(write_csv, "WRITE_SIZE")

    def steps(cnt, enc):
        keys = ("ht_alloc_init_kernel",) if enc else ("ht_dec_vlc_kernel", "t1_dec_kernel")
        n = sum(v for k, v in cnt.items() if any(s in k for s in keys))
        if not enc and n == 0:                   # (a Part-1 frame whose blocks all went to the lane decoder)
            n = sum(v for k, v in cnt.items() if "t1_lanes_kernel" in k)
        return max(1, n)
    out = {}
    for fam, _ in FAMILIES:
        if fam not in f and fam not in w:
            continue
        enc = fam in ("dwt_level0_fused", "dwt_levels_1plus", "ht_encode_kernel", "ingest_kernel")
        nf, nw = steps(fc, enc), steps(wc, enc)
        rd = f.get(fam, 0.0) * 1024 * 2 / nf          # KiB -> B, gfx950 half-count correction
        wr = w.get(fam, 0.0) * 1024 / nw
        out[fam] = {"steps_sampled": nf, "hbm_read_bytes_per_step": int(rd), "hbm_write_bytes_per_step": int(wr),
                    "hbm_bytes_per_step": int(rd + wr)}
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main(*sys.argv[1:4])
