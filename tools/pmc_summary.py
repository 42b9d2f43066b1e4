#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes of tools/prof_pmc_scanline.sh: per kernel, mean counter value per dispatch,
with the corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads (doubled here, flagged); 12-byte-per-lane (dwordx3) accesses are an
uncalibrated width.  usage: pmc_summary.py DIR"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
V = 1080 * 1920 * 192


def short_name(k):
    """'void (anonymous namespace)::k_scan<3, 2, 3, true>((anonymous namespace)::ScanArgs)' -> 'k_scan<3, 2, 3, true>'"""
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return k[:i]
    return k


def per_kernel(pattern, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in counters:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


out = {"scanline_1080p_d192": {}, "ncc_450x375_w21": {}}
f = per_kernel(f"{root}/scan_fetch/**/*counter_collection.csv", {"FETCH_SIZE"})
w = per_kernel(f"{root}/scan_write/**/*counter_collection.csv", {"WRITE_SIZE"})
alg = {"k_scan_lr": 16, "k_scan<3, 2": 16, "k_scan<3, 3": 12}
for k in sorted(set(f) | set(w)):
    if "k_scan" not in k:
        continue
    short = short_name(k)
    fv, wv = f.get(k, {}).get("FETCH_SIZE", []), w.get(k, {}).get("WRITE_SIZE", [])
    fetch = 2.0 * 1024 * sum(fv) / max(1, len(fv))
    write = 1024.0 * sum(wv) / max(1, len(wv))
    a = next((v for n, v in alg.items() if n in k), None)
    out["scanline_1080p_d192"][short] = {
        "dispatches": [len(fv), len(wv)], "fetch_bytes_x2_corrected": round(fetch), "write_bytes": round(write),
        "hbm_bytes_per_hypothesis": round((fetch + write) / V, 2), "algorithmic_bytes_per_hypothesis": a,
        "traffic_over_algorithmic": round((fetch + write) / V / a, 3) if a else None}
for D in (64, 200):
    n = per_kernel(f"{root}/ncc_{D}/**/*counter_collection.csv", {"SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"})
    for k, c in n.items():
        if "k_ncc2" not in k:
            continue
        m = {a: sum(b) / len(b) for a, b in c.items()}
        out["ncc_450x375_w21"][f"D={D}"] = {
            "kernel": short_name(k), **{a: round(b) for a, b in m.items()},
            "lds_bytes_if_all_b32": round(m.get("SQ_INSTS_LDS", 0) * 256),
            "lds_array_busy": round(m.get("SQ_LDS_IDX_ACTIVE", 0) / 256.0 / (m.get("GRBM_GUI_ACTIVE", 1) / 8.0), 3),
            "note": "SQ_INSTS_LDS wave-instructions x 256 B (64 lanes x 4 B): the kernel's LDS traffic is ds_read_b32; "
                    "GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
print(json.dumps(out, indent=1))
