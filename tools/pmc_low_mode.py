#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of tools/hbm_low_mode.sh: per process and engine cycle, the mean duration of the 8M-member
step_kernel dispatches (from the Start / End timestamps rocprofv3 writes beside every counter sample) and the mean of every
collected counter per dispatch, plus the derived per-request figures (EA latency = *_LEVEL / *REQ, stall cycles per request).
    python3 tools/pmc_low_mode.py <pass_dir> <cycles> [min_grid]      -> a table on stdout"""
import csv
import glob
import os
import sys


def main():
    d, cycles = sys.argv[1], int(sys.argv[2])
    min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000
    rows = {}                      # pid -> dispatch id -> {"dur": us, counter: value}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if "fiveeq::step_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) < min_grid:
                    continue
                rec = rows.setdefault(r["Process_Id"], {}).setdefault(int(r["Dispatch_Id"]), {})
                rec["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                rec[r["Counter_Name"]] = float(r["Counter_Value"])
    names = sorted({k for p in rows.values() for rec in p.values() for k in rec} - {"dur"})
    print("pid cycle dispatches dur_us " + " ".join(names))
    for pid, disp in sorted(rows.items()):
        ids = sorted(disp)
        per = len(ids) // cycles
        for c in range(cycles):
            grp = [disp[i] for i in ids[c * per:(c + 1) * per]][per // 4:]          # skip the first quarter: warm-up launches
            if not grp:
                continue
            mean = lambda k: sum(g.get(k, 0.0) for g in grp) / len(grp)              # noqa: E731
            print(pid, c, len(grp), f"{mean('dur'):.1f}", " ".join(f"{mean(k):.4g}" for k in names))


if __name__ == "__main__":
    main()
