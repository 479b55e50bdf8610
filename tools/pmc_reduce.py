#!/usr/bin/env python3
"""Mean per dispatch of every counter rocprofv3 --pmc collected for the fiveeq kernels under the given pass directories.
    python3 tools/pmc_reduce.py <label> <pass_dir> [...]   ->  label, kernel, counter, dispatches, mean, duration_us"""
import collections
import csv
import glob
import os
import sys

label = sys.argv[1]
acc, dur = collections.defaultdict(list), collections.defaultdict(list)
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "fiveeq::" not in name:
                    continue
                k = name.split("(")[0].replace("void ", "").strip()
                acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
                dur[(k, row["Counter_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
for (k, c), v in sorted(acc.items()):
    print(f"{label},\"{k}\",{c},{len(v)},{sum(v) / len(v):.6g},{sum(dur[(k, c)]) / len(v):.1f}")
