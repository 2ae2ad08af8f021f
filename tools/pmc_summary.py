#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files into something small enough to commit and read:
  * a filtered CSV per pass (this library's kernels only, kernel names without their argument lists), and
  * one JSON with the mean counter value per dispatch of every kernel, plus per-wave figures where SQ_WAVES is there.

    python3 tools/pmc_summary.py <out prefix> <pass dir> [<pass dir> ...]
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("vvc355::", "")
    return re.sub(r"\(.*$", "", name)


def main():
    prefix, dirs = sys.argv[1], sys.argv[2:]
    summary = {}
    for d in dirs:
        path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        tag = os.path.basename(os.path.normpath(d))
        rows = [r for r in csv.DictReader(open(path)) if "vvc355::" in r["Kernel_Name"]]
        with open(f"{prefix}_{tag}.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"])
            for r in rows:
                w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"],
                            r["Counter_Name"], r["Counter_Value"]])
        acc = {}
        for r in rows:
            acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            e = summary.setdefault(k, {})
            for c, v in cs.items():
                e[c] = sum(v) / len(v)
                e["dispatches_" + tag] = len(v)
    for k, e in summary.items():
        if e.get("SQ_WAVES"):
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                if c in e:
                    e[c + "_per_wave"] = round(e[c] / e["SQ_WAVES"], 1)
    with open(f"{prefix}_summary.json", "w") as f:
        json.dump({"_note": "mean per dispatch of each kernel; rocprofv3 --pmc passes listed in profiles/README.md; FETCH_SIZE / WRITE_SIZE in KiB "
                            "(read bytes = 2 x FETCH_SIZE x 1024 on gfx950, see tools/pmc_traffic.py)", "kernels": summary}, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
