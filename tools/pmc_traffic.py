#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc CSV passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC has only 4
slots) of `bench.py` into profiles/pmc_traffic.json.

Corrections (MI355X_MICROARCH.md, section HBM):
  - counter unit is KiB-like: bytes = value * 1024 ... on ROCm 7.2 the derived FETCH_SIZE /
    WRITE_SIZE are reported in KB (TCC_EA0_RDREQ*64B/1024 etc.);
  - on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
    -> doubled;
  - WRITE_SIZE is exact for 16-B-per-lane streaming stores; the D=192 kernel stores 12 B per lane
    (dwordx3), which the guide lists as uncalibrated -> reported as is and flagged.
usage: pmc_traffic.py <fetch_csv> <write_csv> <workload> <kernel-substring> <tag> [dispatches-per-pair]
"""
import csv
import json
import os
import sys


def per_kernel(path, counter, needle):
    tot, n = 0.0, 0
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == counter:
                tot += float(row["Counter_Value"])
                n += 1
    return tot, n


def main():
    fetch_csv, write_csv, workload, needle, tag = sys.argv[1:6]
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE", needle)
    w, nw = per_kernel(write_csv, "WRITE_SIZE", needle)
    launches_per_pair = int(sys.argv[6]) if len(sys.argv) > 6 else 1   # dispatches of that kernel per pair
    fetch_b = 2.0 * f * 1024 / (nf / launches_per_pair)    # gfx950: x2
    write_b = w * 1024 / (nw / launches_per_pair)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles", "pmc_traffic.json")
    d = json.load(open(out)) if os.path.exists(out) else {}
    d[workload] = {"hbm_bytes_per_pair": round(fetch_b + write_b), "fetch_bytes_per_pair_x2_corrected": round(fetch_b),
                   "write_bytes_per_pair": round(write_b), "dispatches": [nf, nw], "kernel": needle,
                   "source": f"profiles/{tag}_pmc_fetch.csv + profiles/{tag}_pmc_write.csv (rocprofv3 --pmc, separate passes; "
                             "FETCH_SIZE doubled per the gfx950 note; dwordx3 stores are an uncalibrated width)"}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d[workload], indent=1))


if __name__ == "__main__":
    main()
